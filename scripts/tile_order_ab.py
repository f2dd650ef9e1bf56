#!/usr/bin/env python3
"""VERDICT r3 item 4 -- traffic -> clock, measured once: the bulk update (`syrk_kernel`) and K^^-1 = L^-T L^-1
(`lauum_kernel`) of the headline batched plan (32 sites, n = 8192, fp64) in the default row order against S x S SUPERTILE
orders (DGP_OPT_SYRK_ORDER / DGP_OPT_LAUUM_ORDER), A/B'd by WALL (HIP-event stage times, rounds interleaved in one
process) and by IN-KERNEL CLOCK (dgp_debug_clock_probe while the stage runs back to back, >= 2 s of launches before).
Results must be bitwise equal (same tiles, same sums).  usage: python scripts/tile_order_ab.py [n=8192] [sites=32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from discontinuum_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
plan, th, r, noise = bench.make_plan("loadest", n, 3, torch.float64, dev, S, 1)
plan.set_timing(True)
lib = _lib.load()


def potrf_loop():
    plan.stage_gram(th, noise)
    plan.stage_potrf()


ref = None
variants = [("rows (default)", 0, 0), ("syrk 4x4", 4, 0), ("syrk 8x8", 8, 0), ("lauum 4x4", 0, 4), ("lauum 8x8", 0, 8),
            ("both 8x8", 8, 8), ("rows again", 0, 0)]
for _ in range(6):  # >= 2 s of back-to-back launches before the first reading
    plan.fit_step(th, r, noise)
print(f"{'variant':16s} {'step ms':>8s} {'potrf':>7s} {'bulk':>7s} {'TF/s':>6s} {'lauum':>7s} {'TF/s':>6s} | clock MHz: potrf-loop lauum-loop | bitwise")
for name, so, lo in variants:
    plan.set_option(_lib.OPT_SYRK_ORDER, so)
    plan.set_option(_lib.OPT_LAUUM_ORDER, lo)
    best = None
    for rnd in range(3):
        for _ in range(2):
            out = plan.fit_step(th, r, noise)[0]
        torch.cuda.synchronize()
        ms = plan.get_timing()
        tot = sum(ms[k] for k in (_lib.TIME_GRAM, _lib.TIME_POTRF, _lib.TIME_TRTRI, _lib.TIME_LAUUM, _lib.TIME_SOLVE, _lib.TIME_GRAD))
        if best is None or tot < best[0]:
            best = (tot, list(ms))
    tot, ms = best
    host = out.cpu()
    if ref is None:
        ref = host
    same = bool(torch.equal(host, ref))
    N = plan.N
    ck_p = bench.clock_probe(lib, dev, potrf_loop, (ms[_lib.TIME_GRAM] + ms[_lib.TIME_POTRF]) * 1e-3)
    plan.fit_step(th, r, noise)  # restore T for the lauum loop
    ck_l = bench.clock_probe(lib, dev, plan.stage_lauum, ms[_lib.TIME_LAUUM] * 1e-3)
    print(f"{name:16s} {tot:8.2f} {ms[_lib.TIME_POTRF]:7.2f} {ms[_lib.TIME_SYRK_SUM]:7.2f} {ms[_lib.TIME_SYRK_FLOP] / ms[_lib.TIME_SYRK_SUM] / 1e9:6.1f} "
          f"{ms[_lib.TIME_LAUUM]:7.2f} {S * N ** 3 / 3 / ms[_lib.TIME_LAUUM] / 1e9:6.1f} | {ck_p['mhz'] if ck_p else None!s:>10} {ck_l['mhz'] if ck_l else None!s:>10} | {same}", flush=True)
