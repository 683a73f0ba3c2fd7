"""Driver of tools/glds/probe_glds.hip: global -> LDS streaming rate per address pattern / occupancy (no compute)."""
import ctypes as C
import subprocess
import sys
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
SO = HERE / "libprobe_glds.so"


def build():
    if not SO.exists() or SO.stat().st_mtime < (HERE / "probe_glds.hip").stat().st_mtime:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", str(HERE / "probe_glds.hip"), "-o", str(SO)])
    return C.CDLL(str(SO))


def main():
    lib = build()
    if "--build-only" in sys.argv:
        return
    lib.glds_probe_run.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    sink = torch.zeros(1, dtype=torch.int32, device="cuda")
    if "--strides" in sys.argv:
        buf = torch.zeros(24 << 20, dtype=torch.uint8, device="cuda")
        for ld in (640, 1280, 1408, 2560, 2688, 3840, 5120, 5248, 7680, 10240, 10368, 20480, 20608):
            grid, iters, tiles, stages, lds = 2048, 20, 2, 2, 65536
            def run():
                lib.glds_probe_run(buf.data_ptr(), buf.numel(), ld, 0, iters, tiles, grid, stages, lds, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
            run()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                run()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 5
            total = grid * tiles * iters * 32768
            print(f"rows of 128 B at stride {ld:6d} B (2 wg/CU): {total / ms / 1e9:7.2f} TB/s", flush=True)
        return
    for mb in (24, 512):  # a working set that lives in L2 / Infinity Cache, and one that streams from HBM
        buf = torch.zeros(mb << 20, dtype=torch.uint8, device="cuda")
        for (label, mode, ld) in (("rows of 128 B at stride 2560", 0, 2560), ("rows of 128 B at stride 10240", 0, 10240), ("32 KB contiguous", 1, 0), ("rows of 256 B at stride 2560", 2, 2560)):
            for (stages, lds, occ) in ((2, 65536, "2 wg/CU"), (2, 131072, "1 wg/CU"), (3, 98304, "1 wg/CU, 2 in flight")):
                grid, iters, tiles = 2048, 20, 2
                def run():
                    lib.glds_probe_run(buf.data_ptr(), buf.numel(), ld, mode, iters, tiles, grid, stages, lds, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
                run()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    run()
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b) / 5
                total = grid * tiles * iters * 32768
                print(f"{mb:4d} MB set | {label:32s} | {occ:22s}: {total / ms / 1e9:7.2f} TB/s  ({total / ms / 1e9 * 1e12 / 256 / 2.4e9:5.1f} B/clk/CU)", flush=True)


if __name__ == "__main__":
    main()
