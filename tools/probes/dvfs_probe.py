#!/usr/bin/env python3
"""Is the score kernel limited by the chip's power budget?  Same launch, same instruction stream, different inputs:
random image (the bench workload), constant image (every B operand byte equal), near-constant templates, and an
image with the statistics of a photograph (neighbouring pixels correlated: box-filtered noise at three scales;
templates cut from it).  Kernel time (HIP events) and the shader clock measured inside the kernel.  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multitemplatematching-python_amd"))
import numpy as np
import synth
from MTM import _lib
img, units, _ = synth.make_workload(seed=3, image_hw=(2160, 3840), n_base=32, templ=64, noisy_per_unit=3)
ctx = _lib.Context(0)
def run(name, im, tl):
    ctx.set_image(im); ctx.set_templates(tl, 5)
    for _ in range(300): ctx.find_matches(0, 0.5)
    ks, cl = [], []
    for _ in range(60):
        ctx.find_matches(0, 0.5); t = ctx.timing(); ks.append(t["ncc_kernel_ms"]); cl.append(t["sclk_mhz"])
    print("%-46s kernel %.4f ms  in-kernel clock %.0f MHz  -> %.0f cycles/us x ms = %.3f Mcycles" % (
        name, np.median(ks), np.median(cl), np.median(cl), np.median(ks) * np.median(cl) / 1e3))
tl = [(u[1], None) for u in units]
run("random image, random templates", img, tl)
run("constant image (128), random templates", np.full_like(img, 128), tl)
half = np.full((64, 64), 128, np.uint8); half[::2] = 127          # non-constant (variance > 0) but nearly constant bytes
run("constant image, near-constant templates", np.full_like(img, 128), [(half, None)] * 32)
run("random image, near-constant templates", img, [(half, None)] * 32)


nat = synth.smooth_u8(11, img.shape)
ntl = [(t, None) for _, t in synth.cut_templates(5, nat, 32, 64)]
ctx.set_image(nat); ctx.set_templates(ntl, 5)
# threshold 0.9: few candidates (at 0.5 a smooth image has millions - another regime, see dense_probe.py)
def run_thr(name, thr):
    for _ in range(300): ctx.find_matches(0, thr)
    ks, cl = [], []
    for _ in range(60):
        ctx.find_matches(0, thr); t = ctx.timing(); ks.append(t["ncc_kernel_ms"]); cl.append(t["sclk_mhz"])
    print("%-46s kernel %.4f ms  in-kernel clock %.0f MHz  -> %.3f Mcycles" % (name, np.median(ks), np.median(cl), np.median(ks) * np.median(cl) / 1e3))
run_thr("photograph-like image + crops, threshold 0.9", 0.9)
