#!/usr/bin/env python3
"""Scan the library's gfx950 code objects for stores that sit behind an `s_waitcnt vmcnt(0)`.

Loads and stores share ONE in-order counter on gfx9 (vmcnt): a vmcnt(0) in front of a store waits for every store issued before it.
The compiler places such a wait when a loaded value is first used inside a divergent branch of an unrolled loop (it cannot prove the
wait of the previous iteration was executed) -- the epilogue then pays one store round trip per row.  Found this way in round 5:
conv16's streaming kernel (bias used inside `if (off < 0) continue`), gemm_k's epilogue (all 64 stores).

    python tools/wait_store_scan.py [cl_ica_amd/lib/libclica_hip.so]      (no GPU needed; uses llvm-objcopy / llvm-objdump of /opt/rocm)
prints per kernel: stores behind a vmcnt(0), longest run of consecutive such stores, stores in the kernel.
"""
import os, re, struct, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "cl_ica_amd", "lib", "libclica_hip.so")
tmp = tempfile.mkdtemp()
fat = os.path.join(tmp, "fat.bin")
subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "copy.so")])
blob = open(fat, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
rows = []
for m in re.finditer(re.escape(magic), blob):
    p = m.start()
    n = struct.unpack_from("<Q", blob, p + 24)[0]
    o = p + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", blob, o); o += 24
        triple = blob[o:o + tl].decode(); o += tl
        if "gfx950" not in triple or not size:
            continue
        elf = os.path.join(tmp, "img.elf")
        open(elf, "wb").write(blob[p + off:p + off + size])
        asm = subprocess.run([f"{LLVM}/llvm-objdump", "-d", elf], capture_output=True, text=True).stdout
        name, ops = None, []
        def flush():
            if not name:
                return
            cnt = run = best = stores = 0
            waited = False
            for op in ops:
                if op == "W":
                    waited = True
                else:
                    stores += 1
                    if waited:
                        cnt += 1; run += 1; best = max(best, run)
                    else:
                        run = 0
                    waited = False
            if cnt >= 4:
                rows.append((cnt, best, stores, name))
        for line in asm.splitlines():
            mm = re.match(r"^[0-9a-f]+ <(.*)>:", line)
            if mm:
                flush(); name, ops = mm.group(1), []
            elif "s_waitcnt" in line and "vmcnt(0)" in line:
                ops.append("W")
            elif re.search(r"\t(global_store|buffer_store|flat_store|global_atomic|buffer_atomic)", line):
                ops.append("S")
        flush()
names = "\n".join(r[3] for r in rows)
try:
    dem = subprocess.run(["c++filt"], input=names, capture_output=True, text=True).stdout.splitlines()
except OSError:
    dem = names.splitlines()
print("%8s %8s %8s  kernel" % ("waited", "run", "stores"))
for (cnt, best, stores, _), d in sorted(zip(rows, dem), reverse=True):
    print("%8d %8d %8d  %s" % (cnt, best, stores, d[:140]))
