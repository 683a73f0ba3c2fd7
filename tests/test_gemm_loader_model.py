"""Address-sequence model of gemm.hip's loader, CPU only.  The kernel's K loop was rewritten so that row pointers and K offsets
are hoisted out of it (`enter` / `set_tap` / running offsets); this is a line-by-line Python restatement of BOTH loaders (the
per-iteration one kept behind ABL = 4 / mi355x_set_option("legacy") and the hoisted one) checked against each other on random
plain / conv launches: multi-segment (3x3 + 1x1 + LoRA), stride, nearest-2x, both paddings, K-blocked operands, split-K
entry points and the K-rotation probe.  The kernel itself is checked on the GPU (tests/test_kernels_gpu.py); this guards the
control flow of segment / tap transitions, which those cases exercise only for the shapes they happen to pick."""
import random


class Seg:
    def __init__(s, **kw): s.__dict__.update(kw)

def old_seq(p, first_seg, first_kb, n_issue, xm, xcoff, wnrow, wcoff, xb, xoy, xox, xvalid, rotate):
    out = []; seg, kb = first_seg, first_kb
    for _ in range(n_issue):
        sp = p["seg"][seg]
        dy = dx = 0; cb = kb
        if p["conv"]:
            tap = kb // sp.cpb; cb = kb - tap * sp.cpb
            dy = tap // sp.ksize; dx = tap - dy * sp.ksize; dy -= sp.pad; dx -= sp.pad
        xs = []
        for it in range(len(xm)):
            if p["conv"]:
                iy = xoy[it] * sp.stride + dy; ix = xox[it] * sp.stride + dx
                HH = sp.H << sp.ups; WW = sp.W << sp.ups
                ok = xvalid[it] and 0 <= iy < HH and 0 <= ix < WW
                sy, sx = iy >> sp.ups, ix >> sp.ups
                pix = (xb[it] * sp.H + sy) * sp.W + sx
                xs.append(("x", seg, pix * sp.ldxb + cb * 128 + xcoff[it]) if ok else ("z", xcoff[it]))
            else:
                xs.append(("x", seg, (kb * p["M"] + xm[it]) * 128 + xcoff[it]) if sp.xkb else ("x", seg, xm[it] * sp.ldxb + kb * 128 + xcoff[it]))
        ws = [("w", seg, (kb * p["N"] + wnrow[it]) * 128 + wcoff[it]) if sp.wkb else ("w", seg, wnrow[it] * sp.ldwb + kb * 128 + wcoff[it]) for it in range(len(wnrow))]
        out.append((xs, ws))
        kb += 1
        if kb == sp.nkb:
            kb = 0
            if not rotate: seg += 1
    return out

def new_seq(p, first_seg, first_kb, n_issue, xm, xcoff, wnrow, wcoff, xb, xoy, xox, xvalid, rotate):
    out = []; st = dict(seg=first_seg, kb=first_kb)
    def set_tap(sp):
        dy = st["tap"] // sp.ksize; dx = st["tap"] - dy * sp.ksize; dy -= sp.pad; dx -= sp.pad
        st["xbase"] = []
        for it in range(len(xm)):
            iy = xoy[it] * sp.stride + dy; ix = xox[it] * sp.stride + dx
            HH = sp.H << sp.ups; WW = sp.W << sp.ups
            ok = xvalid[it] and 0 <= iy < HH and 0 <= ix < WW
            sy, sx = iy >> sp.ups, ix >> sp.ups
            pix = (xb[it] * sp.H + sy) * sp.W + sx
            st["xbase"].append(pix * sp.ldxb + xcoff[it] if ok else None)
    def enter(s, kb0):
        sp = p["seg"][s]
        st["cur_nkb"], st["cur_cpb"] = sp.nkb, sp.cpb
        st["wstep"] = p["N"] * 128 if sp.wkb else 128
        st["woff"] = kb0 * st["wstep"]
        st["wbase"] = [(wnrow[it] * 128 if sp.wkb else wnrow[it] * sp.ldwb) + wcoff[it] for it in range(len(wnrow))]
        if p["conv"]:
            st["tap"] = kb0 // sp.cpb; st["cb"] = kb0 - st["tap"] * sp.cpb; set_tap(sp)
        else:
            st["xstep"] = p["M"] * 128 if sp.xkb else 128
            st["xoff"] = kb0 * st["xstep"]
            st["xbase"] = [(xm[it] * 128 if sp.xkb else xm[it] * sp.ldxb) + xcoff[it] for it in range(len(xm))]
    enter(st["seg"], st["kb"])
    for _ in range(n_issue):
        seg = st["seg"]
        if p["conv"]:
            xs = [("x", seg, b + st["cb"] * 128) if b is not None else ("z", xcoff[it]) for it, b in enumerate(st["xbase"])]
        else:
            xs = [("x", seg, b + st["xoff"]) for b in st["xbase"]]
        ws = [("w", seg, b + st["woff"]) for b in st["wbase"]]
        out.append((xs, ws))
        st["kb"] += 1; st["woff"] += st["wstep"]
        if p["conv"]:
            st["cb"] += 1
            if st["cb"] == st["cur_cpb"]:
                st["cb"] = 0; st["tap"] += 1
                if st["kb"] < st["cur_nkb"]: set_tap(p["seg"][st["seg"]])
        else:
            st["xoff"] += st["xstep"]
        if st["kb"] == st["cur_nkb"]:
            st["kb"] = 0
            if not rotate: st["seg"] += 1
            if st["seg"] < len(p["seg"]): enter(st["seg"], 0)
    return out

def test_hoisted_loader_visits_the_same_addresses_as_the_per_iteration_loader():
  random.seed(0)
  for trial in range(1500):
      conv = random.random() < 0.5
      nseg = random.randint(1, 3)
      M, N = random.randint(1, 400), random.randint(1, 300)
      segs = []
      OH, OW, B = random.randint(1, 9), random.randint(1, 9), random.randint(1, 2)
      for s in range(nseg):
          if conv:
              ksize = random.choice([1, 3]); stride = random.choice([1, 2]); ups = random.choice([0, 1]); cpb = random.randint(1, 4)
              H = max(1, (OH * stride) >> ups) + random.randint(0, 1); W = max(1, (OW * stride) >> ups) + random.randint(0, 1)
              segs.append(Seg(nkb=ksize * ksize * cpb, cpb=cpb, ksize=ksize, stride=stride, ups=ups, H=H, W=W, pad=random.choice([0, ksize // 2]), ldxb=cpb * 128 + random.choice([0, 256]),
                              ldwb=ksize * ksize * cpb * 128, wkb=random.random() < 0.5, xkb=False))
          else:
              nkb = random.randint(1, 6)
              segs.append(Seg(nkb=nkb, cpb=1, ksize=1, stride=1, ups=0, H=0, W=0, pad=0, ldxb=nkb * 128 + random.choice([0, 128]), ldwb=nkb * 128 + random.choice([0, 128]),
                              wkb=random.random() < 0.5, xkb=random.random() < 0.3))
      if conv: M = B * OH * OW
      p = dict(conv=conv, M=M, N=N, seg=segs)
      total = sum(s.nkb for s in segs)
      rotate = nseg == 1 and random.random() < 0.2
      ksplit = 1 if rotate else random.choice([1, 1, 2, 3])
      XI = WI = 2
      rows = [random.randint(0, max(M - 1, 0) + 3) for _ in range(XI)]
      xvalid = [r < M for r in rows]; xm = [min(r, M - 1) for r in rows]
      xb = [m // (OH * OW) for m in xm]; rem = [m - b * OH * OW for m, b in zip(xm, xb)]
      xoy = [r // OW for r in rem]; xox = [r - y * OW for r, y in zip(rem, xoy)]
      xcoff = [random.randrange(8) * 16 for _ in range(XI)]; wcoff = [random.randrange(8) * 16 for _ in range(WI)]
      wnrow = [random.randint(0, N - 1) for _ in range(WI)]
      for split in range(ksplit):
          seg, kb, n_issue = 0, 0, total
          if rotate: kb = random.randrange(total)
          if ksplit > 1:
              per = -(-total // ksplit); first = split * per; n_issue = min(per, total - first)
              if n_issue <= 0: continue
              kb = first
              while seg < nseg - 1 and kb >= segs[seg].nkb: kb -= segs[seg].nkb; seg += 1
          a = old_seq(p, seg, kb, n_issue, xm, xcoff, wnrow, wcoff, xb, xoy, xox, xvalid, rotate)
          b = new_seq(p, seg, kb, n_issue, xm, xcoff, wnrow, wcoff, xb, xoy, xox, xvalid, rotate)
          assert a == b, (trial, conv, nseg, rotate, ksplit, split)

