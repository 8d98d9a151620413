#!/usr/bin/env python3
"""Registers, scratch, LDS and kernel-argument bytes of every gfx950 kernel in a library / object (no GPU needed).
    python tools/kernel_resources.py [lib-or-object] [name filter] [--dump-elf DIR]
reads the AMDGPU metadata note of the embedded code objects (llvm-objcopy + llvm-readelf of /opt/rocm)."""
import os, re, struct, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
lib = args[0] if args else os.path.join(os.path.dirname(__file__), "..", "cl_ica_amd", "lib", "libclica_hip.so")
flt = args[1] if len(args) > 1 else ""
dump = sys.argv[sys.argv.index("--dump-elf") + 1] if "--dump-elf" in sys.argv else None
tmp = tempfile.mkdtemp()
fat = os.path.join(tmp, "fat.bin")
subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "copy")])
blob = open(fat, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
k = 0
for m in re.finditer(re.escape(magic), blob):
    p = m.start()
    n = struct.unpack_from("<Q", blob, p + 24)[0]
    o = p + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", blob, o); o += 24
        triple = blob[o:o + tl].decode(); o += tl
        if "gfx950" not in triple or not size:
            continue
        elf = os.path.join(dump or tmp, f"img{k}.elf"); k += 1
        open(elf, "wb").write(blob[p + off:p + off + size])
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", elf], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            f = lambda key: (re.search(r"\." + key + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", f("name")], capture_output=True, text=True).stdout.strip()
            if flt and flt not in name:
                continue
            print(f"{name[:110]:110s} vgpr {f('vgpr_count'):>4} agpr {blk.split()[0]:>3} sgpr {f('sgpr_count'):>4} spill_v {f('vgpr_spill_count'):>4} "
                  f"scratch {f('private_segment_fixed_size'):>5} lds {f('group_segment_fixed_size'):>6} kernarg {f('kernarg_segment_size'):>5}")
