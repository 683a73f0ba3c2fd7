"""Golden tables for Euler / DPM-Solver++: timesteps, sigmas, model-input scales and the end point of a
trajectory driven by seeded random "noise predictions", from the REAL reference's solvers.
Run in the build container only:  python oracle/make_golden_solvers.py"""
from __future__ import annotations

import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "oracle" / "shim"), "/root/reference/src", str(ROOT)]

import torch  # noqa: E402

from refiners.foundationals.latent_diffusion.solvers import DPMSolver, Euler, LCMSolver  # noqa: E402


def main() -> None:
    out = {}
    for name, n, kw, seed in (("euler_10", 10, {}, 1), ("euler_30", 30, {}, 2), ("dpm_10", 10, {}, 3), ("dpm_30", 30, {}, 4), ("dpm_25_last_first_order", 25, {"last_step_first_order": True}, 5)):
        solver = Euler(n) if name.startswith("euler") else DPMSolver(n, **kw)
        g = torch.Generator().manual_seed(seed)
        x = torch.randn((1, 4, 8, 8), generator=g)
        eps = [torch.randn((1, 4, 8, 8), generator=g) for _ in range(n)]
        scales = [float(solver.scale_model_input(torch.ones(1), s)) for s in range(n)]
        for s in range(n):
            x = solver(x, eps[s], s)
        out[name] = {"solver": name.split("_")[0], "steps": n, "seed": seed, "last_step_first_order": bool(kw.get("last_step_first_order", False)),
                     "timesteps": solver.timesteps.tolist(), "sigmas": solver.sigmas.double().tolist(), "input_scales": scales,
                     "final": x.double().reshape(-1).tolist()}
    for name, n, orig, seed in (("lcm_4", 4, 50, 6), ("lcm_8_of_40", 8, 40, 7)):  # stochastic: the noise generator is part of the recipe
        solver = LCMSolver(n, num_orig_steps=orig)
        g = torch.Generator().manual_seed(seed)
        x = torch.randn((1, 4, 8, 8), generator=g)
        eps = [torch.randn((1, 4, 8, 8), generator=g) for _ in range(n)]
        gn = torch.Generator().manual_seed(1000 + seed)
        for s in range(n):
            x = solver(x, eps[s], s, generator=gn)
        out[name] = {"solver": "lcm", "steps": n, "orig_steps": orig, "seed": seed, "noise_seed": 1000 + seed, "timesteps": solver.timesteps.tolist(),
                     "timestep_indices": [int(i) for i in solver.timestep_indices], "dpm_timesteps": solver.dpm.timesteps.tolist(),
                     "final": x.double().reshape(-1).tolist()}
    (ROOT / "tests" / "golden" / "solvers.json").write_text(json.dumps(out))
    print({k: v["timesteps"][:3] for k, v in out.items()})


if __name__ == "__main__":
    main()
