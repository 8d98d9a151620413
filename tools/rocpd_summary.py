#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (``--kernel-trace --stats``) as a per-kernel table
(calls, total/avg/min/max duration, % of GPU time) -- the text committed under profiles/."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {path}",
             f"{'calls':>8} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel"]
    for name, cnt, tot, avg, mn, mx in rows:
        short = name if len(name) < 150 else name[:147] + "..."
        lines.append(f"{cnt:8d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100.0 * tot / total:6.2f}  {short}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
