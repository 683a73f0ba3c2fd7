"""Round profile on the GPU box: rocprofv3 kernel trace of the timed steps of bench.py (the final build, no extras) + separate PMC
passes (HBM-side traffic, MFMA utilisation, SQ wait / issue breakdown), folded into the files that get committed under profiles/:

  <tag>_bench_rocprof_kernel_stats.md   per-kernel table of the traced run (tools/rocpd_stats.py)
  <tag>_inplace_by_shape.md/.json       every launch of one step mapped back to its program entry: time per (entry point, shape) IN PLACE
  <tag>_pmc_traffic.json                FETCH_SIZE / WRITE_SIZE per family (what bench.py's roofline.traffic reads)
  <tag>_pmc_mfma.json                   SQ_VALU_MFMA_BUSY_CYCLES vs SQ_BUSY_CYCLES per family (roofline.mfma_util)
  <tag>_pmc_sq.json                     SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY / LDS conflicts per family

    python tools/profile_round.py --tag r02_d [--workload lora_ip] [--skip-pmc]
Counter passes use --kernel-trace only (never combined with sys / hip / hsa tracing)."""
from __future__ import annotations

import argparse
import glob
import json
import os
import re
import sqlite3
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "gpurun_out"


def run(cmd: list[str], log: Path) -> int:
    env = dict(os.environ, TMPDIR="/tmp")
    with open(log, "w") as f:
        return subprocess.call(cmd, stdout=f, stderr=subprocess.STDOUT, cwd="/tmp", env=env)


def find_db(d: Path) -> str | None:
    dbs = sorted(glob.glob(str(d / "**" / "*.db"), recursive=True), key=os.path.getmtime)
    return dbs[-1] if dbs else None


def family(kernel: str) -> str | None:
    m = re.search(r"gemm_kernelI\w+?Li\d+ELi\d+ELi\d+ELi\d+ELb([01])E", kernel)
    if m:
        return "mi355x_gemm(conv)" if m.group(1) == "1" else "mi355x_gemm"
    m = re.search(r"gemm8_kernelI\w+?Lb([01])E", kernel)  # the 8-wave / eight-phase loop (csrc/gemm8_kernel.cuh)
    if m:
        return "mi355x_gemm(conv)" if m.group(1) == "1" else "mi355x_gemm"
    if "splitk_reduce_kernel" in kernel:
        return "mi355x_gemm(conv)"
    if re.search(r"attn_(pipe_|short_|general_)?kernel", kernel):
        return "mi355x_attention"
    if "layernorm_kernel" in kernel:
        return "mi355x_layernorm"
    if re.search(r"gn_(partial|finalize|finalize_cs|apply|fused)_kernel", kernel):
        return "mi355x_groupnorm"
    return None


def columns(con: sqlite3.Connection, table: str) -> list[str]:
    return [r[1] for r in con.execute(f"pragma table_info({table})")]


def kernel_rows(db: str) -> list[tuple[str, int, int]]:
    con = sqlite3.connect(db)
    cols = columns(con, "kernels")
    start = "start" if "start" in cols else ("start_timestamp" if "start_timestamp" in cols else None)
    if start is None:
        raise RuntimeError(f"kernels view has columns {cols}")
    return [(n, int(s), int(d)) for n, s, d in con.execute(f"select name, {start}, duration from kernels order by {start}")]


def is_ours(n: str) -> bool:
    return "mi355x" in n or family(n) is not None or ("_kernel" in n and "at::" not in n)


def step_map(rows: list[tuple], program: list[dict], max_steps: int = 4) -> list[list[tuple[dict, list]]]:
    """rows: (kernel name, start, payload) of every dispatch of the traced process, in start order.  Returns, for up to `max_steps` of
    the LAST full replays of the step program, the program entries paired with the payloads of the dispatches they issued (a split-K conv
    = 2 dispatches, a GroupNorm = 3, or 2 when its statistics come from the launch that produced its input) -- i.e. the prologue, the warm-up and every torch kernel are left out.  Steps are delimited by the
    CFG + solver kernel that closes each of them."""
    ours = [(n, s, x) for n, s, x in rows if is_ours(n)]
    ends = [i for i, (n, _, _) in enumerate(ours) if "cfg_ddim_kernel" in n or "cfg_linear_step_kernel" in n]
    if len(ends) < 3:
        return []
    period = ends[-1] - ends[-2]
    out = []
    for e in ends[-max_steps:]:
        if e - period + 1 < 0:
            continue
        st = ours[e - period + 1 : e + 1]
        i, ok, mapped = 0, True, []
        for ent in program:
            what = ent["what"]
            if i >= len(st):
                ok = False
                break
            n = st[i][0]
            take = 1
            if what.startswith("mi355x_gemm"):
                if "gemm_kernel" not in n and "gemm8_kernel" not in n:
                    ok = False
                    break
                if i + 1 < len(st) and "splitk_reduce_kernel" in st[i + 1][0] and ent.get("ksplit", 1) > 1:
                    take = 2
            elif what == "mi355x_groupnorm":
                take = 2 if "gn_finalize_cs_kernel" in n else 3  # (statistics from the producer's epilogue: finalize, apply) / partial sums, finalize, apply
            mapped.append((ent, [x for _, _, x in st[i : i + take]]))
            i += take
        if ok:
            out.append(mapped)
    return out


def inplace(db: str, program: list[dict], tag: str) -> None:
    steps = step_map([(n, s, d) for n, s, d in kernel_rows(db)], program)
    if not steps:
        print("in-place mapping: program / trace mismatch (or fewer than three step boundaries)", file=sys.stderr)
        return
    per_class: dict[str, list[float]] = {}
    for mapped in steps:
        acc: dict[str, float] = {}
        for ent, durs in mapped:
            acc[ent["key"]] = acc.get(ent["key"], 0.0) + sum(durs)
        for k, v in acc.items():
            per_class.setdefault(k, []).append(v)
    used_steps = len(steps)
    counts: dict[str, int] = {}
    for ent in program:
        counts[ent["key"]] = counts.get(ent["key"], 0) + 1
    table = sorted(((k, sum(v) / len(v) / 1e6, counts[k]) for k, v in per_class.items()), key=lambda r: -r[1])
    total = sum(r[1] for r in table)
    lines = [f"in-place time per (entry point, shape) class, mean of {used_steps} traced steps; sum of kernel durations {total:.3f} ms per step", "",
             "| class | launches | ms per step | us per launch | % |", "|---|---|---|---|---|"]
    for k, ms, n in table:
        lines.append(f"| `{k}` | {n} | {ms:.4f} | {ms / n * 1e3:.2f} | {100 * ms / total:.1f} |")
    (OUT / f"{tag}_inplace_by_shape.md").write_text("\n".join(lines) + "\n")
    (OUT / f"{tag}_inplace_by_shape.json").write_text(json.dumps({"steps": used_steps, "sum_ms": total, "classes": [{"class": k, "launches": n, "ms": ms} for k, ms, n in table]}, indent=1))
    print("\n".join(lines[:24]))


def pmc_families(db: str, program: list[dict] | None = None) -> dict:
    """Counter sums per kernel family -- and, with the recorded step program, per shape class -- over the dispatches of the STEP PROGRAM
    only (the last full replays of it in the traced process: no prologue, no warm-up, no set-up kernels).  Without a program (the
    calibration launch) every dispatch of the process counts.  Returns {"scope", "families": {family: {counter: {dispatches, sum}}},
    "classes": {class key: {counter: {dispatches, sum, launches}}}}."""
    con = sqlite3.connect(db)
    # every dispatch of the traced process comes from the kernel trace (a filtered counter pass -- --kernel-include-regex -- holds counters for some kernels
    # only, and the mapping onto the recorded program needs them all); counters are joined on the dispatch's start timestamp (one pmc_events row per XCD
    # instance of a dispatch: summed)
    pm: dict[int, dict] = {}
    for did, name, start, counter, val in con.execute("select dispatch_id, name, min(start), counter_name, sum(counter_value) from pmc_events group by dispatch_id, counter_name"):
        pm.setdefault(start, {})[counter] = val
    rows = []
    for n, st_, dur in kernel_rows(db):
        cs = dict(pm.get(st_, {}), _name=n)
        if st_ in pm:
            cs["DURATION_NS"] = float(dur)  # the dispatch's own duration from the kernel trace of the same run: a pseudo counter (counted dispatches only)
        rows.append((n, st_, cs))
    fams: dict = {}
    classes: dict = {}
    kernels: dict = {}

    def add(store: dict, key: str, cs: dict, launches: int = 0) -> None:
        for counter, val in cs.items():
            if counter == "_name":
                continue
            e = store.setdefault(key, {}).setdefault(counter, {"dispatches": 0, "sum": 0.0, "launches": 0})
            e["dispatches"] += 1
            e["sum"] += val
            e["launches"] += launches

    def short(n: str) -> str:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        return n if len(n) < 120 else n[:117] + "..."

    steps = step_map(rows, program) if program else []
    if steps:
        scope = f"step program only: {len(steps)} full replay(s) of the {len(program)} recorded launches"
        for mapped in steps:
            for ent, payloads in mapped:
                f = {"mi355x_gemm(conv)": "mi355x_gemm(conv)"}.get(ent["what"], ent["what"])
                for j, cs in enumerate(payloads):
                    add(fams, f, cs)
                    add(classes, ent["key"], cs, launches=1 if j == 0 else 0)
                    add(kernels, short(cs["_name"]), cs)
    else:
        scope = "whole process (no step program given, or it did not match the trace)"
        for name, _, cs in rows:
            f = family(name)
            if f is not None:
                add(fams, f, cs)
    return {"scope": scope, "families": fams, "classes": classes, "kernels": kernels}


def traffic_micro(tag: str, prof: Path) -> None:
    """FETCH_SIZE / WRITE_SIZE of the dominant shape classes from tools/probe_traffic.py (a few dozen dispatches instead of a whole bench run): the
    fallback when the counter passes over the step die in the profiler.  Writes <tag>_pmc_traffic.json in the schema bench.py reads."""
    res: dict[str, list] = {}
    classes = None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = prof / f"micro_{counter}"
        for attempt in range(3):
            rc = run(["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", str(d), "--", sys.executable, str(ROOT / "tools" / "probe_traffic.py")], prof / f"micro_{counter}.log")
            if rc == 0:
                break
        print("traffic micro-program", counter, "rc", rc)
        db = find_db(d)
        if rc != 0 or not db:
            return
        for ln in (prof / f"micro_{counter}.log").read_text(errors="replace").splitlines():
            if ln.startswith("TRAFFIC_PROGRAM "):
                classes = json.loads(ln[len("TRAFFIC_PROGRAM "):])
        con = sqlite3.connect(db)
        per: dict[int, list] = {}
        for did, name, start, val in con.execute("select dispatch_id, name, min(start), sum(counter_value) from pmc_events where counter_name = ? group by dispatch_id", (counter,)):
            per[did] = [name, start, val]
        rows = [r for r in sorted(per.values(), key=lambda r: r[1]) if "gemm_kernel" in r[0] or "gemm8_kernel" in r[0]]
        res[counter] = rows
    if not classes or sum(c["launches"] for c in classes) != len(res["FETCH_SIZE"]) or len(res["FETCH_SIZE"]) != len(res["WRITE_SIZE"]):
        print("traffic micro-program: dispatch count mismatch", None if not classes else sum(c["launches"] for c in classes), {k: len(v) for k, v in res.items()})
        return
    out_cls, i = {}, 0
    fam = {"mi355x_gemm": {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]}, "mi355x_gemm(conv)": {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]}}
    for c in classes:
        n = c["launches"]
        f = sum(r[2] for r in res["FETCH_SIZE"][i : i + n]) / n * 1024 * 2  # KiB, doubled (MI355X_MICROARCH.md, HBM section)
        w = sum(r[2] for r in res["WRITE_SIZE"][i : i + n]) / n * 1024
        i += n
        out_cls[c["class"]] = {"FETCH_SIZE": {"launches": n, "bytes_per_launch": f}, "WRITE_SIZE": {"launches": n, "bytes_per_launch": w}, "algorithmic_bytes": c["algorithmic_bytes"],
                               "fetched_plus_written_over_algorithmic": (f + w) / c["algorithmic_bytes"], "launches_per_step": c["per_step"]}
        fk = "mi355x_gemm(conv)" if c["class"].startswith("conv") else "mi355x_gemm"
        for counter, v in (("FETCH_SIZE", f), ("WRITE_SIZE", w)):
            fam[fk][counter][0] += v * c["per_step"]
            fam[fk][counter][1] += c["per_step"]
    fams = {fk: {counter: {"dispatches": tot[1], "bytes_per_launch": tot[0] / max(tot[1], 1)} for counter, tot in cs.items()} for fk, cs in fam.items()}
    (OUT / f"{tag}_pmc_traffic.json").write_text(json.dumps({
        "how": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python tools/probe_traffic.py (one pass per counter): the dominant shape classes of the step, "
               "6 launches each on fresh operands, outside the step (cold operands: an upper bound for the in-place traffic)",
        "scope": "micro-program per shape class (the counter passes over the whole step died in the profiler); families = the classes weighted by their launches per step",
        "units": "counter values are KiB; FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section)", "families": fams, "classes": out_cls}, indent=1))
    print({k: round(v["fetched_plus_written_over_algorithmic"], 2) for k, v in out_cls.items()})


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--workload", default="lora_ip")
    ap.add_argument("--skip-pmc", action="store_true")
    ap.add_argument("--lora-mode", default="fused")
    ap.add_argument("--groups", default="", help="skip the unfiltered attempt of every PMC pass and collect these kernel-name groups only (4wave,8wave,other)")
    ap.add_argument("--passes", default="fetch,write,mfma,sq", help="which PMC passes to run (a pass the profiler crashed in can be repeated alone)")
    args = ap.parse_args()
    OUT.mkdir(exist_ok=True)
    tag = args.tag
    prof = Path("/tmp") / f"{tag}_prof"  # raw profiler databases stay on the box: only the summaries go to gpurun_out/ (64 MiB cap)
    prof.mkdir(exist_ok=True)
    bench = [sys.executable, str(ROOT / "bench.py"), "--workload", args.workload, "--no-cpu-baseline", "--no-extra", "--no-roofline", "--no-graph", "--lora-mode", args.lora_mode]
    prog_path = prof / "program.json"
    # 1. kernel trace of the timed steps
    rc = run(["rocprofv3", "--kernel-trace", "--stats", "-d", str(prof / "kt"), "--", *bench, "--steps", "5", "--warmup", "2", "--dump-program", str(prog_path)], prof / "kt.log")
    db = find_db(prof / "kt")
    print("kernel trace rc", rc, "db", db)
    if db:
        subprocess.call([sys.executable, str(ROOT / "tools" / "rocpd_stats.py"), db, str(OUT / f"{tag}_bench_rocprof_kernel_stats.md")], stdout=subprocess.DEVNULL)
        if prog_path.exists():
            inplace(db, json.loads(prog_path.read_text()), tag)
    if args.skip_pmc:
        return
    how = "rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --workload %s --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-roofline --no-graph (one pass per counter set)" % args.workload
    passes = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"], "mfma": ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"],
              "sq": ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES"]}
    got: dict[str, dict] = {}
    got_cls: dict[str, dict] = {}
    got_k: dict[str, dict] = {}
    scope = None
    program = json.loads(prog_path.read_text()) if prog_path.exists() else None
    passes = {k: v for k, v in passes.items() if k in args.passes.split(",")}
    # Kernel-name groups for the fallback: the counter collection of this pool's profiler build dies (SIGSEGV inside the dispatch interception) on some
    # whole-step passes -- rounds 4 and 5 lost every FETCH_SIZE / WRITE_SIZE pass that way.  A pass that dies unfiltered is repeated as three
    # processes, each collecting counters for ONE group of kernels (--kernel-include-regex / --kernel-exclude-regex; the kernel trace still holds every
    # dispatch, so the mapping onto the recorded program works per process) and the per-class sums are merged.
    groups = [("4wave", ["--kernel-include-regex", "gemm_kernel|splitk_reduce_kernel"]), ("8wave", ["--kernel-include-regex", "gemm8_kernel"]),
              ("other", ["--kernel-exclude-regex", "gemm_kernel|gemm8_kernel|splitk_reduce_kernel"])]
    if args.groups:
        groups = [g for g in groups if g[0] in args.groups.split(",")]
    how_groups: dict[str, str] = {}

    def merge(dst: dict, src: dict) -> None:
        for key, cs in src.items():
            for counter, e in cs.items():
                d = dst.setdefault(key, {}).setdefault(counter, {"dispatches": 0, "sum": 0.0, "launches": 0})
                for f in ("dispatches", "sum", "launches"):
                    d[f] += e[f]

    def one_run(name: str, counters: list, extra: list, sub: str) -> dict | None:
        d = prof / f"{name}{sub}"
        rc = -1
        for attempt in range(2):  # the collection dies with SIGSEGV at the first dispatch now and then on this pool: try again
            rc = run(["rocprofv3", "--pmc", *counters, "--kernel-trace", *extra, "-d", str(d), "--", *bench, "--steps", "2", "--warmup", "1"], prof / f"{name}{sub}.log")
            if rc == 0:
                break
            print("pmc pass", name + sub, "attempt", attempt + 1, "rc", rc)
        db = find_db(d)
        print("pmc pass", name + sub, "rc", rc, "db", db)
        if rc != 0:  # keep what the profiler said: the raw logs stay on the box
            tail = [ln for ln in (prof / f"{name}{sub}.log").read_text(errors="replace").splitlines() if "amdgpu.ids" not in ln][-40:]
            (OUT / f"{tag}_pmc_{name}{sub}_failed.log").write_text("\n".join(tail) + "\n")
            return None
        try:
            return pmc_families(db, program) if db else None
        except Exception as exc:  # noqa: BLE001
            print("  failed to read", exc)
            return None

    for name, counters in passes.items():
        r = None if args.groups else one_run(name, counters, [], "")
        if r is not None and "step program only" in r["scope"]:
            how_groups[name] = "one unfiltered process"
        else:
            r = {"scope": None, "families": {}, "classes": {}, "kernels": {}}
            done, died = [], []
            for gname, extra in groups:
                rg = one_run(name, counters, extra, "_" + gname)
                if rg is None or "step program only" not in rg["scope"]:
                    died.append(gname)
                    continue
                done.append(gname)
                r["scope"] = rg["scope"]
                for part in ("families", "classes", "kernels"):
                    merge(r[part], rg[part])
            how_groups[name] = f"kernel-name groups, one process each: collected {done or 'none'}" + (f", died in the profiler: {died}" if died else "")
            if not done:
                r = None
            else:
                r["scope"] += f"; counters collected per kernel-name group in separate processes ({', '.join(done)}" + (f"; NOT {', '.join(died)}: the profiler died" if died else "") + ")"
        if r is not None:
            got[name], got_cls[name], scope = r["families"], r["classes"], r["scope"]
            got_k[name] = r["kernels"]
            if name == "sq":  # which kernel VARIANT loses LDS cycles to bank conflicts (the family figure hides it)
                by_k = {k: {c: v["sum"] for c, v in cs.items()} for k, cs in r["kernels"].items()}
                for k, cs in by_k.items():
                    if cs.get("SQ_LDS_IDX_ACTIVE"):
                        cs["lds_conflict_frac"] = cs.get("SQ_LDS_BANK_CONFLICT", 0.0) / cs["SQ_LDS_IDX_ACTIVE"]
                (OUT / f"{tag}_pmc_sq_by_kernel.json").write_text(json.dumps({"how": how, "scope": scope, "kernels": by_k}, indent=1))
            print("  scope:", scope)
    how = how + " | " + "; ".join(f"{k}: {v}" for k, v in how_groups.items())
    if "fetch" in got and "write" in got:
        fams = {}
        for f in set(got["fetch"]) | set(got["write"]):
            fams[f] = {}
            for counter, src in (("FETCH_SIZE", got["fetch"]), ("WRITE_SIZE", got["write"])):
                d = src.get(f, {}).get(counter)
                if d:
                    calls = d["dispatches"]
                    kb = d["sum"] / calls
                    fams[f][counter] = {"dispatches": calls, "raw_kb_per_launch": kb, "bytes_per_launch": kb * 1024 * (2 if counter == "FETCH_SIZE" else 1)}
        cls = {}
        for k in set(got_cls.get("fetch", {})) | set(got_cls.get("write", {})):
            row = {}
            for counter, src in (("FETCH_SIZE", got_cls.get("fetch", {})), ("WRITE_SIZE", got_cls.get("write", {}))):
                d = src.get(k, {}).get(counter)
                if d and d["launches"]:
                    row[counter] = {"launches": d["launches"], "bytes_per_launch": d["sum"] / d["launches"] * 1024 * (2 if counter == "FETCH_SIZE" else 1)}
            cls[k] = row
        # the HBM-bound kernels of the step in GB/s: per kernel name, fetched (doubled) + written bytes over the dispatches' own durations
        hbm = {}
        for k in set(got_k.get("fetch", {})) & set(got_k.get("write", {})):
            if "gemm" in k or "attn" in k:
                continue
            fk, wk = got_k["fetch"][k], got_k["write"][k]
            if "FETCH_SIZE" in fk and "WRITE_SIZE" in wk and fk.get("DURATION_NS", {}).get("sum"):
                n = fk["FETCH_SIZE"]["dispatches"]
                byts = fk["FETCH_SIZE"]["sum"] * 1024 * 2 + wk["WRITE_SIZE"]["sum"] * 1024 * n / max(wk["WRITE_SIZE"]["dispatches"], 1)
                ns = fk["DURATION_NS"]["sum"]
                hbm[k] = {"dispatches": n, "bytes_per_dispatch": byts / n, "avg_us": ns / n / 1e3, "gbps": byts / ns, "frac_of_8TBps": byts / ns / 8000.0}
        (OUT / f"{tag}_pmc_traffic.json").write_text(json.dumps({"how": how, "scope": scope, "hbm_bound_kernels": hbm, "units": "counter values are KiB; FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section); families: per KERNEL dispatch (GroupNorm: per kernel, not per call); classes: per program launch (all its kernels)", "families": fams, "classes": cls}, indent=1))
        print({f: {c: round(v["bytes_per_launch"] / 1e6, 2) for c, v in cs.items()} for f, cs in fams.items()})
    if not ("fetch" in got and "write" in got) and ("fetch" in passes or "write" in passes):
        traffic_micro(tag, prof)
    # calibration of the MFMA-busy counter on a launch whose MFMA count is known exactly (tools/probe_mfma_cal.py)
    cal = None
    rc = run(["rocprofv3", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "--kernel-trace", "-d", str(prof / "cal"), "--", sys.executable, str(ROOT / "tools" / "probe_mfma_cal.py")], prof / "cal.log")
    db = find_db(prof / "cal")
    if db:
        try:
            c = pmc_families(db)["families"].get("mi355x_gemm", {})
            n_launch = c["SQ_VALU_MFMA_BUSY_CYCLES"]["dispatches"]
            busy = c["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / n_launch
            gui = c["GRBM_GUI_ACTIVE"]["sum"] / n_launch
            n_mfma = 4096 ** 3 * 2 / 16384
            con = sqlite3.connect(db)
            dur = con.execute("select avg(duration) from kernels where name like '%gemm_kernel%'").fetchone()[0]
            cal = {"launch": "4096^3 bf16, 128x128 tile", "mfma_instructions": n_mfma, "avg_duration_us": dur / 1e3, "tflops": 2 * 4096 ** 3 / (dur * 1e-9) / 1e12,
                   "SQ_VALU_MFMA_BUSY_CYCLES_over_GRBM_GUI_ACTIVE": busy / gui}
            cal["mfma_util_by_construction"] = cal["tflops"] / 2500.0  # known MFMA count, known duration
            cal["SQ_VALU_MFMA_BUSY_CYCLES_per_ns"] = busy / dur
            cal["note"] = ("pmc_events holds one row per XCD instance of a dispatch and GRBM_GUI_ACTIVE does not tick at the shader clock on this profiler build: "
                           "only the RATIO of the two counters is used, anchored on this launch")
            print("mfma counter calibration:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in cal.items()})
        except Exception as exc:  # noqa: BLE001
            print("mfma calibration failed:", exc)
    if "mfma" in got:
        fams = {}
        for f, cs in got["mfma"].items():
            busy, sq, waves = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("sum"), cs.get("SQ_BUSY_CYCLES", {}).get("sum"), cs.get("SQ_WAVE_CYCLES", {}).get("sum")
            gui = cs.get("GRBM_GUI_ACTIVE", {}).get("sum")
            row = {"dispatches": next(iter(cs.values()))["dispatches"], "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_BUSY_CYCLES": sq, "SQ_WAVE_CYCLES": waves, "GRBM_GUI_ACTIVE": gui}
            if busy and gui:
                row["busy_over_gui_active"] = busy / gui
                if cal:  # fraction of GPU-active time with the matrix pipes busy, anchored on the calibration launch's known utilisation
                    row["mfma_util"] = cal["mfma_util_by_construction"] * row["busy_over_gui_active"] / cal["SQ_VALU_MFMA_BUSY_CYCLES_over_GRBM_GUI_ACTIVE"]
            if busy and sq:
                row["mfma_busy_over_sq_busy"] = busy / sq
            ns = cs.get("DURATION_NS", {}).get("sum")
            if busy and ns and cal:
                # the same anchoring against the dispatches' OWN durations: GRBM_GUI_ACTIVE also ticks through the profiler's per-dispatch counter
                # start / stop (about 7-15 us per dispatch here), which dilutes 20-40 us launches far more than the 132 us calibration launch
                row["DURATION_NS"] = ns
                row["mfma_util_by_duration"] = cal["mfma_util_by_construction"] * (busy / ns) / cal["SQ_VALU_MFMA_BUSY_CYCLES_per_ns"]
            fams[f] = row
        cls = {}
        for k, cs in got_cls.get("mfma", {}).items():  # per shape class: the same anchored ratio
            busy, gui = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("sum"), cs.get("GRBM_GUI_ACTIVE", {}).get("sum")
            if busy and gui and cal:
                cls[k] = {"launches": cs["GRBM_GUI_ACTIVE"]["launches"], "mfma_util": cal["mfma_util_by_construction"] * (busy / gui) / cal["SQ_VALU_MFMA_BUSY_CYCLES_over_GRBM_GUI_ACTIVE"]}
                ns = cs.get("DURATION_NS", {}).get("sum")
                if ns:
                    cls[k]["avg_us_under_pmc"] = ns / 1e3 / max(cs["GRBM_GUI_ACTIVE"]["launches"], 1)
                    cls[k]["mfma_util_by_duration"] = cal["mfma_util_by_construction"] * (busy / ns) / cal["SQ_VALU_MFMA_BUSY_CYCLES_per_ns"]
        (OUT / f"{tag}_pmc_mfma.json").write_text(json.dumps({"how": how, "scope": scope, "classes": cls, "note": "mfma_util = (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE of the family) / (the same ratio of the calibration launch) x the calibration launch's known utilisation", "calibration": cal, "families": fams}, indent=1))
        print({f: {k: (round(v, 4) if isinstance(v, float) and v < 10 else v) for k, v in r.items()} for f, r in fams.items()})
    if "sq" in got:
        fams = {}
        for f, cs in got["sq"].items():
            w = cs.get("SQ_WAVE_CYCLES", {}).get("sum") or 0
            row = {c: d["sum"] for c, d in cs.items()}
            if w:
                for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                    if c in cs:
                        row[c + "_frac_of_wave_cycles"] = cs[c]["sum"] / w
            fams[f] = row
        (OUT / f"{tag}_pmc_sq.json").write_text(json.dumps({"how": how, "scope": scope, "families": fams}, indent=1))
        print({f: {k: round(v, 3) for k, v in r.items() if k.endswith("frac_of_wave_cycles")} for f, r in fams.items()})


if __name__ == "__main__":
    main()
