"""Static dispatch arithmetic of the step's GEMM / conv launches (CPU only; no GPU, no torch): for every (entry point, shape) class of an in-place
table (profiles/r0N_*_inplace_by_shape.json, written by tools/profile_round.py on the GPU box) --

  tile        the tile the launch runs on (the class name carries a tuned choice; otherwise the library heuristic, pick_tile in csrc/gemm_kernel.cuh)
  head        workgroups ahead of the output tiles that occupy resident slots without producing output: LoRA producers (ceil(M / 32) x groups) or
              t-tiles (ceil(M / BM)), rounded up to 8 like launch_cfg does  (weight-prefetch workgroups, 32 ... 128 per launch and gone after a few us,
              are not in the class name and are left out)
  slots       resident workgroups of that kernel instantiation on the chip: 256 CUs x min(register limit, LDS limit), from the compiler's own resource
              report (hipcc -Rpass-analysis=kernel-resource-usage, cached as JSON beside the table) and the LDS formula of launch_cfg
  rounds      (head + tiles) / slots: 1.03 means 3 % of the workgroups wait a full workgroup-duration for a slot (what 192 producers did to the
              480 tiles of Q|K|V^T before the t-tiles)
  per CU / balance   output tiles per CU and (tiles / 256) / ceil(tiles / 256): 640 tiles = 2.5 per CU = half the CUs run 3 tiles while the other
              half run 2 -- the launch takes 3 tile-durations for 2.5 tile-durations of work (balance 0.83), whatever the slots allow
The head / tile rules applied are those of the CURRENT library (t-tiles for multi-group launches on a wide enough tile), not necessarily those of
the build the table was traced on.

Round 4 found two such cases by hand (Q|K|V^T's producers; the 4096-token Q|K|V^T on a tile too narrow for t-tiles).  This prints them all.

  python tools/dispatch_report.py profiles/r04_g_inplace_by_shape.json [--resources profiles/r04_q_kernel_resources.json] [--md out.md]"""
from __future__ import annotations

import json
import math
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "refiners_amd" / "csrc"
TILES = {1: (128, 128), 2: (128, 64), 3: (64, 128), 4: (64, 64), 6: (128, 128)}
CUS, LDS_CU, VGPR_SIMD = 256, 160 * 1024, 512
G8_TILES = {7: (256, 256), 8: (256, 256), 9: (192, 256)}  # 8 = stream-K over the same tiles (every CU gets an equal share of (tiles x K tiles): balance 1 by construction)


def kernel_resources(cache: Path) -> dict:
    """{"<T>,<BM>,<BN>,<CONV>,<NSTAGE>,<KG>,<LORA>": {"vgpr": .., "agpr": .., "occupancy": ..}} from the compiler's remarks on gemm.hip / gemm_conv.hip."""
    if cache.exists():
        return json.loads(cache.read_text())
    out = {}
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                               "-Rpass-analysis=kernel-resource-usage", "-c", str(CSRC / src), "-o", "/dev/null"], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)
             for src in ("gemm.hip", "gemm_conv.hip")]
    for p in procs:
        _, err = p.communicate()
        for block in re.split(r"remark: [^\n]*Function Name: ", err)[1:]:
            name = block.split()[0]
            m = re.match(r"_ZN6mi355x11gemm_kernelI(f|DF16b)Li(\d+)ELi(\d+)ELi2ELi2ELb([01])ELi(\d+)ELi(\d+)ELb([01])EEEvNS_5GemmPE", name)
            if not m:
                continue
            def field(key):
                g = re.search(re.escape(key) + r": (\d+)", block)
                return int(g.group(1)) if g else -1
            key = ",".join(("f32" if m.group(1) == "f" else "bf16",) + m.groups()[1:])
            out[key] = {"vgpr": field("VGPRs"), "agpr": field("AGPRs"), "spill": field("VGPRs Spill"), "occupancy": field("Occupancy [waves/SIMD]")}
    cache.write_text(json.dumps(out, indent=1, sort_keys=True))
    return out


def pick_tile(M: int, N: int, conv: bool, geglu: bool) -> int:
    b128 = -(-M // 128) * -(-N // 128)
    if conv:
        return 3
    if geglu:
        return 1
    return 4 if b128 <= 256 else 2 if b128 < 1000 else 1


def analyse(cls: str, res: dict) -> dict | None:
    parts = cls.split(":")
    if parts[0] not in ("gemm", "conv") or len(parts) < 5:
        return None
    conv, dt = parts[0] == "conv", parts[1]
    M, N, K = (int(v) for v in parts[2].split("x"))
    flags = parts[4]
    lora, geglu, ln = flags.endswith("lora"), "geglu" in flags, "ln" in flags
    groups = 3 if (lora and re.search(r"T\d+", flags)) else 1
    R = 32  # (the bench workload: two rank-16 adapters stacked)
    tile, stages, tuned = 0, 2, False
    if len(parts) > 5 and parts[5].startswith("tile"):
        tile, stages = (int(v) for v in parts[5][4:].split("/"))
        stages = stages or 2  # (0 = the library's default depth)
        tuned = True
    if tile == 0:
        tile = pick_tile(M, N, conv, geglu)
    if geglu and tile in (2, 4):
        tile = 3
    if lora:
        stages = 2
        if conv and tile not in (1, 3):
            tile = 3
    if tile in G8_TILES:  # the 8-wave loop (csrc/gemm8_kernel.cuh): one workgroup per CU by construction (256 registers x 8 waves, 122-138 KB of LDS); its in-launch LoRA has t-tiles
        BM, BN = G8_TILES[tile]
        tiles = -(-M // BM) * -(-N // BN)
        head, kind = (-(-(-(-M // BM)) // 8) * 8, "t-tiles") if lora else (0, "")
        rounds = (tiles + head) / CUS
        return {"class": cls, "tile": f"{BM}x{BN}" + ("*" if tuned else "") + (" stream-K" if tile == 8 else ""), "stages": 2, "vgpr": 256 if tile != 9 else 224, "lds_kb": round((2 * (BM + BN) * 128 + 10240) / 1024, 1),
                "per_cu": 1, "tiles": tiles, "head": head, "head_kind": kind, "slots": CUS, "rounds": round(rounds, 3), "tiles_per_cu": tiles / CUS,
                "balance": 1.0 if tile == 8 else tiles / CUS / math.ceil(tiles / CUS)}
    BM, BN = TILES[tile]
    KG = 2 if tile == 6 else 1
    key = f"{dt},{BM},{BN},{int(conv)},{stages},{KG},{int(lora)}"
    r = res.get(key)
    if r is None:
        return None
    es = 4 if dt == "f32" else 2
    lds = KG * stages * (BM + BN) * 128 + (BM * 8 if ln else 0) + (BN * 32 * es if lora else 0)
    waves = 4 * KG
    by_regs = min(8, VGPR_SIMD // max(8, -(-(r["vgpr"] + max(r["agpr"], 0)) // 8) * 8)) * 4 // waves
    per_cu = max(1, min(by_regs, LDS_CU // lds))
    tiles = -(-M // BM) * -(-N // BN)
    head, kind = 0, ""
    if lora:
        if not conv and groups * R <= BN and groups > 1:
            head, kind = -(-(-(-M // BM)) // 8) * 8, "t-tiles"
        else:
            head, kind = -(-(-(-M // 32) * groups) // 8) * 8, "producers"
    wgs, slots = tiles + head, per_cu * CUS
    rounds = wgs / slots
    return {"class": cls, "tile": f"{BM}x{BN}" + ("*" if tuned else ""), "stages": stages, "vgpr": r["vgpr"], "lds_kb": round(lds / 1024, 1), "per_cu": per_cu, "tiles": tiles,
            "head": head, "head_kind": kind, "slots": slots, "rounds": round(rounds, 3), "tiles_per_cu": tiles / CUS, "balance": tiles / CUS / math.ceil(tiles / CUS)}


def main() -> None:
    table = Path(sys.argv[1])
    cache = Path(sys.argv[sys.argv.index("--resources") + 1]) if "--resources" in sys.argv else table.with_name(table.name.split("_inplace")[0] + "_kernel_resources.json")
    res = kernel_resources(cache)
    data = json.loads(table.read_text())
    rows = []
    for c in data["classes"]:
        a = analyse(c["class"], res)
        if a:
            a.update(launches=c["launches"], ms=c["ms"], us=c["ms"] / c["launches"] * 1e3)
            rows.append(a)
    lines = [f"dispatch arithmetic of `{table.name}` (tools/dispatch_report.py; `*` = tuned tile; slots = 256 CUs x workgroups per CU of that instantiation)", "",
             "| class | launches | ms/step | us | tile | VGPRs | LDS KB | resident per CU | tiles | head | slots | rounds | tiles per CU | balance |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for a in sorted(rows, key=lambda a: -a["ms"]):
        head = f"{a['head']} {a['head_kind']}" if a["head"] else "--"
        lines.append(f"| `{a['class']}` | {a['launches']} | {a['ms']:.3f} | {a['us']:.1f} | {a['tile']} | {a['vgpr']} | {a['lds_kb']} | {a['per_cu']} | {a['tiles']} | {head} | {a['slots']} | {a['rounds']:.2f} | "
                     f"{a['tiles_per_cu']:.2f} | {a['balance']:.2f} |")
    lost = sum(a["ms"] * (1 - a["balance"]) for a in rows)
    lines += ["", f"sum over classes of ms x (1 - balance) = {lost:.2f} ms of {sum(a['ms'] for a in rows):.2f} ms: the time the step's GEMM / conv launches spend with part of the chip idle "
              "behind the last tile round, IF every tile of a launch took the same time and the dispatcher spread them evenly (an upper bound on what balanced tile shapes could return)"]
    text = "\n".join(lines) + "\n"
    if "--md" in sys.argv:
        Path(sys.argv[sys.argv.index("--md") + 1]).write_text(text)
    print(text)


if __name__ == "__main__":
    main()
