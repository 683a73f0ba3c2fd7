"""Summarise a rocprofv3 rocpd database (the default --kernel-trace --stats output of ROCm 7.2) as a per-kernel table:
calls, total / average / min / max duration, share of GPU time.  Usage: python tools/rocpd_stats.py <results.db> [out.md]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) < 150 else name[:147] + "..."


def main() -> None:
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, lo, hi in rows:
        lines.append(f"| `{short(name)}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {lo / 1e3:.2f} | {hi / 1e3:.2f} | {100 * tot / total:.1f} |")
    text = f"GPU kernel time total: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n\n" + "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
