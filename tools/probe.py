"""First-contact diagnostics on a GPU box: run every kernel parity case (both tile loaders), time the hot shapes,
and dump everything to gpurun_out/probe.json.  Never raises: every failure is recorded and the run continues."""
import json
import os
import sys
import time
import traceback
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from refiners_amd import native  # noqa: E402
from tests import kernel_cases  # noqa: E402

OUT = ROOT / "gpurun_out"
OUT.mkdir(exist_ok=True)


def run_cases(tag):
    rows = []
    for name, thunk in kernel_cases.all_cases():
        t0 = time.time()
        try:
            err, scale, tol = thunk()
            torch.cuda.synchronize()
            ok = err <= tol * scale + 1e-7
            rows.append(dict(name=name, loader=tag, ok=bool(ok), err=err, scale=scale, tol=tol, sec=time.time() - t0))
        except Exception as e:  # noqa: BLE001
            rows.append(dict(name=name, loader=tag, ok=False, error=f"{type(e).__name__}: {e}", tb=traceback.format_exc()[-1500:]))
            try:
                torch.cuda.synchronize()
            except Exception as e2:  # noqa: BLE001
                rows.append(dict(name="SYNC_AFTER_" + name, ok=False, error=str(e2)))
                break
        r = rows[-1]
        print(f"[{tag}] {r['name']:45s} {'OK ' if r['ok'] else 'BAD'} err={r.get('err', float('nan')):.3e} scale={r.get('scale', 0):.2e} {r.get('error', '')}", flush=True)
    return rows


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def perf():
    rows = []
    dev = "cuda"
    for dt, tag in ((torch.bfloat16, "bf16"), (torch.float32, "f32")):
        for (M, K, N) in ((2048, 1280, 1280), (8192, 640, 640), (2048, 1280, 10240), (2048, 5120, 1280), (8192, 2560, 640)):
            x = torch.randn(M, K, device=dev).to(dt)
            w = torch.randn(N, K, device=dev).to(dt)
            o = torch.empty(M, N, device=dev, dtype=dt)
            for glds in (True, False):
                native.set_glds(glds)
                try:
                    t = timeit(lambda: native.gemm([(x, w)], o))
                    rows.append(dict(kind="gemm", dtype=tag, M=M, K=K, N=N, glds=glds, sec=t, tflops=2 * M * K * N / t / 1e12))
                except Exception as e:  # noqa: BLE001
                    rows.append(dict(kind="gemm", dtype=tag, M=M, K=K, N=N, glds=glds, error=str(e)))
            native.set_glds(True)
            t = timeit(lambda: torch.matmul(x, w.t()))
            rows.append(dict(kind="torch_matmul", dtype=tag, M=M, K=K, N=N, sec=t, tflops=2 * M * K * N / t / 1e12))
        # conv 3x3
        for (B, C, Co, H) in ((2, 1280, 1280, 32), (2, 320, 320, 128), (2, 640, 640, 64)):
            x = torch.randn(B, H, H, C, device=dev).to(dt)
            w = torch.randn(Co, 9 * C, device=dev).to(dt)
            o = torch.empty(B * H * H, Co, device=dev, dtype=dt)
            try:
                t = timeit(lambda: native.conv_gemm([(x, w, 3, 1, 1)], o, B, H, H), iters=10)
                rows.append(dict(kind="conv3x3", dtype=tag, B=B, C=C, Co=Co, H=H, sec=t, tflops=2 * B * H * H * 9 * C * Co / t / 1e12))
            except Exception as e:  # noqa: BLE001
                rows.append(dict(kind="conv3x3", dtype=tag, C=C, error=str(e)))
        # attention
        for (B, H, L, Lk) in ((2, 20, 1024, 1024), (2, 10, 4096, 4096), (2, 20, 1024, 77)):
            Cc = H * 64
            q = torch.randn(B, L, Cc, device=dev).to(dt)
            k = torch.randn(B, Lk, Cc, device=dev).to(dt)
            Lkp = (Lk + 63) // 64 * 64
            vt = torch.randn(Cc, B, Lkp, device=dev).to(dt)
            o = torch.empty_like(q)
            for glds in (True, False):
                native.set_glds(glds)
                try:
                    t = timeit(lambda: native.attention(q, o, H, [(k, vt, Lk, 1.0)]), iters=10)
                    rows.append(dict(kind="attn", dtype=tag, B=B, H=H, Lq=L, Lk=Lk, glds=glds, sec=t, tflops=4 * B * L * Lk * Cc / t / 1e12))
                except Exception as e:  # noqa: BLE001
                    rows.append(dict(kind="attn", dtype=tag, Lq=L, Lk=Lk, glds=glds, error=str(e)))
            native.set_glds(True)
            if Lk == L:
                qh = q.reshape(B, L, H, 64).transpose(1, 2)
                t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qh, qh, qh), iters=10)
                rows.append(dict(kind="torch_sdpa", dtype=tag, B=B, H=H, Lq=L, Lk=Lk, sec=t, tflops=4 * B * L * Lk * Cc / t / 1e12))
        # norms
        x = torch.randn(2, 16384, 320, device=dev).to(dt)
        g = torch.ones(320, device=dev, dtype=dt)
        o = torch.empty_like(x)
        t = timeit(lambda: native.groupnorm_nhwc(x, g, g, 32, 1e-5, True, o))
        rows.append(dict(kind="groupnorm", dtype=tag, shape=[2, 16384, 320], sec=t, gbps=3 * x.numel() * x.element_size() / t / 1e9))
        x = torch.randn(2048, 1280, device=dev).to(dt)
        g = torch.ones(1280, device=dev, dtype=dt)
        o = torch.empty_like(x)
        t = timeit(lambda: native.layernorm(x, g, g, 1e-5, o))
        rows.append(dict(kind="layernorm", dtype=tag, shape=[2048, 1280], sec=t, gbps=2 * x.numel() * x.element_size() / t / 1e9))
    for r in rows:
        print(r, flush=True)
    return rows


def main():
    res = {"device": None, "cases": [], "perf": []}
    try:
        res["device"] = native.device_info()
        print(res["device"], flush=True)
        native.set_glds(True)
        res["cases"] += run_cases("glds")
        native.set_glds(False)
        res["cases"] += run_cases("regs")
        native.set_glds(True)
        if "--no-perf" not in sys.argv:
            res["perf"] = perf()
    except Exception as e:  # noqa: BLE001
        res["fatal"] = f"{type(e).__name__}: {e}\n{traceback.format_exc()}"
        print(res["fatal"], flush=True)
    (OUT / "probe.json").write_text(json.dumps(res, indent=1))
    bad = [r for r in res["cases"] if not r.get("ok")]
    print(f"SUMMARY: {len(res['cases']) - len(bad)} ok, {len(bad)} bad", flush=True)
    for r in bad:
        print("  BAD", r.get("loader"), r["name"], r.get("err"), r.get("error", ""), flush=True)


if __name__ == "__main__":
    main()
