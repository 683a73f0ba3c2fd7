"""Fold two per-kernel PMC tables (tools/rocpd_pmc.py output of a `--pmc FETCH_SIZE` pass and of a `--pmc WRITE_SIZE` pass) into
the per-entry-point families bench.py reports.  Usage: python tools/pmc_families.py FETCH.json WRITE.json out.json "<how>"
Counter values are KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for wide coalesced reads on gfx950."""
import json
import re
import sys


def family(kernel: str):
    m = re.search(r"gemm_kernelI\w+?Li\d+ELi\d+ELi\d+ELi\d+ELb([01])E", kernel)
    if m:
        return "mi355x_gemm(conv)" if m.group(1) == "1" else "mi355x_gemm"
    if re.search(r"attn_(pipe_|short_|general_)?kernel", kernel):
        return "mi355x_attention"
    if "layernorm_kernel" in kernel:
        return "mi355x_layernorm"
    if re.search(r"gn_(partial|finalize|apply)_kernel", kernel):
        return "mi355x_groupnorm"
    return None


def main():
    fetch, write = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
    fams = {}
    for counter, rows in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        for r in rows:
            f = family(r["kernel"])
            if f is None or r["counter"] != counter:
                continue
            d = fams.setdefault(f, {}).setdefault(counter, {"dispatches": 0, "raw_kb": 0.0})
            d["dispatches"] += r["dispatches"]
            d["raw_kb"] += r["sum"]
    out = {"how": sys.argv[4] if len(sys.argv) > 4 else "", "units": "counter values are KiB; FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section)", "families": {}}
    for f, cs in fams.items():
        out["families"][f] = {}
        for counter, d in cs.items():
            # GroupNorm is three kernels per entry-point call: report per call
            calls = d["dispatches"] / 3 if f == "mi355x_groupnorm" else d["dispatches"]
            kb = d["raw_kb"] / calls
            out["families"][f][counter] = {"dispatches": d["dispatches"], "raw_kb_per_launch": kb, "bytes_per_launch": kb * 1024 * (2 if counter == "FETCH_SIZE" else 1)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for f, cs in out["families"].items():
        print(f, {c: round(v["bytes_per_launch"] / 1e6, 2) for c, v in cs.items()}, "MB per launch")


if __name__ == "__main__":
    main()
