"""Euler and DPM-Solver++: the mirror's schedules and updates against tables written by the REAL
reference's solvers (tests/golden/solvers.json, oracle/make_golden_solvers.py), and the eight-coefficient linear form the
MI355X step kernel evaluates against the same trajectories."""
import json

import pytest
import torch

from refiners_amd.latent_diffusion.solvers import DPMSolver, Euler, LCMSolver
from tests import support as S

GOLD = json.loads((S.GOLD / "solvers.json").read_text())


def _eps(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn((1, 4, 8, 8), generator=g), [torch.randn((1, 4, 8, 8), generator=g) for _ in range(n)]


def _check_lcm(g):
    """LCMSolver (solvers/lcm.py): schedule, the stochastic trajectory with the reference's own generator protocol, and the
    linear form the step kernel evaluates (history slot = the step's noise draw)."""
    n = g["steps"]
    m = LCMSolver(n, num_orig_steps=g["orig_steps"])
    assert m.timesteps.tolist() == g["timesteps"] and [int(i) for i in m.timestep_indices] == g["timestep_indices"]
    assert m.dpm.timesteps.tolist() == g["dpm_timesteps"]
    x, eps = _eps(n, g["seed"])
    gn, gl = torch.Generator().manual_seed(g["noise_seed"]), torch.Generator().manual_seed(g["noise_seed"])
    xm, xl = x.clone(), x.clone().double()
    for s in range(n):
        xm = m(xm, eps[s], s, generator=gn)
        hx, he, kx, ke, kd, kp, sn = m.linear_step(s)
        hist = torch.randn(eps[s].shape, generator=gl).double() if m.needs_noise(s) else torch.zeros_like(xl)
        d = hx * xl + he * eps[s].double()
        xl = kx * xl + ke * eps[s].double() + kd * d + kp * hist
        assert sn == 1.0
    want = torch.tensor(g["final"], dtype=torch.float64).reshape(1, 4, 8, 8)
    assert torch.equal(xm.double(), want), "mirror LCM solver is not bit-identical to the reference's"
    assert (xl - want).abs().max() / want.abs().max() < 2e-6


@pytest.mark.parametrize("case", sorted(GOLD))
def test_solver_mirror_and_linear_form_match_reference(case):
    g = GOLD[case]
    n = g["steps"]
    if g["solver"] == "lcm":
        return _check_lcm(g)
    make = (lambda: Euler(n)) if g["solver"] == "euler" else (lambda: DPMSolver(n, last_step_first_order=g["last_step_first_order"]))
    m = make()
    assert m.timesteps.tolist() == g["timesteps"]
    assert torch.allclose(m.sigmas.double(), torch.tensor(g["sigmas"], dtype=torch.float64), rtol=0, atol=0)
    x, eps = _eps(n, g["seed"])
    xm, xl, hist = x.clone(), x.clone().double(), torch.zeros_like(x).double()
    for s in range(n):
        if g["solver"] == "euler":
            assert abs(float(m.scale_model_input(torch.ones(1), s)) - g["input_scales"][s]) < 1e-7 and abs(m.input_scale(s) - g["input_scales"][s]) < 1e-6
        xm = m(xm, eps[s], s)
        hx, he, kx, ke, kd, kp, sn = m.linear_step(s)
        d = hx * xl + he * eps[s].double()
        xl, hist = kx * xl + ke * eps[s].double() + kd * d + kp * hist, d
    want = torch.tensor(g["final"], dtype=torch.float64).reshape(1, 4, 8, 8)
    assert torch.equal(xm.double(), want), "mirror solver is not bit-identical to the reference's"
    assert (xl - want).abs().max() / want.abs().max() < 2e-6
