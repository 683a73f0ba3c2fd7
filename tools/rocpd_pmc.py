"""Aggregate a rocprofv3 --pmc run (rocpd database) per kernel name: dispatches, summed counter value, per-dispatch
average and average duration.  Usage: python tools/rocpd_pmc.py <results.db> <out.json>"""
import json
import re
import sqlite3
import sys


def main() -> None:
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    rows = cur.execute(
        "select name, counter_name, count(*), sum(counter_value), avg(counter_value), avg(duration) from pmc_events group by name, counter_name"
    ).fetchall()
    out = []
    for name, counter, n, total, avg, dur in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        out.append({"kernel": short[:160], "counter": counter, "dispatches": n, "sum": total, "avg_per_dispatch": avg, "avg_duration_us": dur / 1e3})
    out.sort(key=lambda r: -r["sum"])
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    for r in out[:14]:
        print(f"{r['counter']:12s} {r['dispatches']:6d} x  avg {r['avg_per_dispatch']:14.1f}  sum {r['sum']:16.1f}  {r['avg_duration_us']:8.1f} us  {r['kernel'][:90]}")


if __name__ == "__main__":
    main()
