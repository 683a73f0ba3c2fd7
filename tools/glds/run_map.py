"""Driver of tools/glds/probe_map.hip: the global -> LDS streaming rate of gfx950 as a function of waves per CU, KB per stage, stages in flight,
barrier, load path and where the data is served from (L2 / Infinity Cache / HBM).  `python tools/glds/run_map.py [--build-only]`"""
import ctypes as C
import subprocess
import sys
from pathlib import Path

if "--build-only" not in sys.argv:
    import torch  # BEFORE the probe library: its HIP runtime must be the one torch has already initialised (a second copy finds no device)

HERE = Path(__file__).resolve().parent
SO = HERE / "libprobe_map.so"


def build():
    if not SO.exists() or SO.stat().st_mtime < (HERE / "probe_map.hip").stat().st_mtime:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", str(HERE / "probe_map.hip"), "-o", str(SO)])
    return C.CDLL(str(SO))


def main():
    lib = build()
    if "--build-only" in sys.argv:
        return
    lib.glds_map_run.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    sink = torch.zeros(1, dtype=torch.int32, device="cuda")
    buf = torch.randn(1 << 30, dtype=torch.bfloat16, device="cuda").view(torch.uint8)  # 2 GB
    MODE = {0: "lds-dma", 1: "to-vgpr", 2: "vgpr+ds_write"}
    print("set/XCD   path           waves/CU  wg/CU  S(KB/stage/wave)  D(in flight)  bar |   TB/s   B/clk/CU(2.4GHz)  KB in flight/CU", flush=True)
    for set_mb, label in ((1, "L2"), (24, "MALL"), (256, "HBM")):
        set_bytes = set_mb << 20
        for (threads, wgcu) in ((256, 1), (256, 2), (512, 1), (256, 4), (512, 2)):
            for (S, D, bar, mode) in ((4, 1, 1, 0), (4, 2, 1, 0), (4, 3, 1, 0), (8, 1, 1, 0), (8, 2, 1, 0), (8, 3, 1, 0), (2, 1, 1, 0), (2, 3, 1, 0), (2, 7, 1, 0),
                                      (4, 1, 0, 0), (4, 2, 0, 0), (4, 3, 0, 0), (8, 1, 0, 0), (8, 2, 0, 0),
                                      (4, 1, 0, 1), (4, 2, 0, 1), (8, 1, 0, 1), (4, 1, 1, 2), (4, 2, 1, 2)):
                nw = threads // 64
                need = nw * (D + 1) * S * 1024
                lds = max(need, 160 * 1024 // wgcu - 1024 if wgcu > 1 else 160 * 1024)
                if need > lds or lds > 160 * 1024:
                    continue
                if label != "L2" and (mode == 2 or (bar == 0 and mode == 0)):
                    continue
                grid = 256 * wgcu
                total_kb_per_wave = 4096  # 4 MB per wave
                iters = total_kb_per_wave // S

                def go():
                    return lib.glds_map_run(buf.data_ptr(), set_bytes, S, D, bar, mode, iters, grid, threads, lds, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)

                if go() != 0:
                    print("launch refused", S, D, bar, mode, threads, lds)
                    continue
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3):
                    go()
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b) / 3
                total = grid * nw * iters * S * 1024
                tbs = total / ms / 1e9
                print(f"{set_mb:4d}MB {label:5s} {MODE[mode]:14s} {nw * wgcu:8d} {wgcu:6d} {S:12d} {D:14d} {bar:8d} | {tbs:7.2f} {tbs * 1e12 / 256 / 2.4e9:10.1f} {nw * wgcu * S * D:18d}", flush=True)


if __name__ == "__main__":
    main()
