#!/usr/bin/env python3
"""Fit the dispatcher's cost model (csrc/fa_fwd16_w64.hip: fwd_w64_predict_us / fwd_16_predict_us) to measured launches.

    python tools/fit_route_model.py [profiles/r5/routing_random_*.jsonl ...]

Input: the records of tools/lab/routing_random_probe.py (per random launch size: the dispatcher's choice, the one-workgroup-per-CU
kernel forced, the 128-row kernel forced; graph-replayed microseconds).  Output: the two constant tables in C++ syntax, each model's
residuals, and what routing by the model costs against the faster of the two measured times -- next to what the dispatcher that took
the measurements cost on the same records.  The model FORMS here and in the C++ must stay in step."""
import glob
import json
import math
import re
import sys
from pathlib import Path

import numpy as np
from scipy.optimize import least_squares

ROOT = Path(__file__).resolve().parent.parent
CUS = 256


def cdiv(a, b):
    return -(-a // b)


def load(paths):
    recs = []
    for f in paths:
        for line in open(f):
            d = json.loads(line)
            if "shape" not in d or "w64_us" not in d or "r128_us" not in d:
                continue
            # a forced kernel that cannot run the shape falls back to the other: not a measurement of it
            if "w64" not in d.get("w64_kernel", "") or "w64" in d.get("r128_kernel", ""):
                continue
            m = re.match(r"B(\d+) H(\d+) Sq(\d+) Skv(\d+) D(\d+) (\w+)", d["shape"])
            B, H, Sq, Skv, D = map(int, m.groups()[:5])
            recs.append(dict(B=B, H=H, Sq=Sq, Skv=Skv, D=D, causal=m.group(6) == "causal", fp16="fp16" in d["w64_kernel"],
                             w64=d["w64_us"], r128=d["r128_us"], dflt=d["default_us"]))
    return recs


def w64_grid(items, T, total):
    if items < CUS and CUS // items >= 2 and items * (CUS // items) <= total:
        return items * (CUS // items)
    if items < CUS and 2 * T * (CUS - items) < 35 * CUS:
        return items
    return min(total, CUS)


def w64_features(r):
    B, H, Sq, Skv, D = r["B"], r["H"], r["Sq"], r["Skv"], r["D"]
    nqb, T = cdiv(Sq, 256), cdiv(Skv, 64)
    items = B * H * nqb
    f = dict(cast=0 if r["fp16"] else 1, vmb=B * H * Skv * D * 2 / 1e6, fold=0.0)
    if r["causal"]:
        jobs = B * H * cdiv(nqb, 2)
        G = min(jobs, CUS)
        rounds = cdiv(jobs, G)
        longest = max(min(T, 4 * j + 4) + (min(T, 4 * (nqb - 1 - j) + 4) if nqb - 1 - j != j else 0) for j in range(cdiv(nqb, 2)))
        f.update(segs=2 * rounds, steps=rounds * longest)
    else:
        G = w64_grid(items, T, items * T)
        full, rem = items // G, items % G
        f.update(segs=full + (1 if rem else 0), steps=full * T + (cdiv(rem * T, G) if rem else 0), fold=max(G / rem - 1, 0) if rem else 0.0)
    return f


def w64_predict(p, f):
    t0, c_seg, t_step, c_fold, cast_a, cast_tbps = p
    return t0 + f["cast"] * (cast_a + f["vmb"] * 2 / cast_tbps) + f["segs"] * c_seg + f["steps"] * t_step + f["fold"] * c_fold


def split_parts(items, nqb, T):  # fa_fwd_16.hip fwd_16_split_plan
    if items > CUS:
        return 1
    k, best = 1, 1e30
    for kk in range(1, (32 if nqb == 1 else 8) + 1):
        if kk > 1 and kk > T // 4:
            break
        w = cdiv(items * kk, CUS)
        rounds = cdiv(w, 2)
        last = w - (rounds - 1) * 2
        cost = T / kk * ((rounds - 1) * 1.25 + (1.25 if last == 2 else 1.0)) + (2.0 * kk if kk > 1 else 0)  # (round 6: 2 tiles per part)
        if cost < best * 0.97:
            best, k = cost, kk
    return k


def r128_features(r):
    B, H, Sq, Skv, D = r["B"], r["H"], r["Sq"], r["Skv"], r["D"]
    nqb, T = cdiv(Sq, 128), cdiv(Skv, 64)
    items = B * H * nqb
    vbytes = B * H * Skv * D * 2
    f = dict(cast=1 if (not r["fp16"] and vbytes >= (16 << 20) and Sq >= 1024) else 0, vmb=vbytes / 1e6,
             R=3 if (D == 128 and not r["causal"]) else 2, causal=r["causal"], fp16=r["fp16"])
    if r["causal"]:
        lens = [min(T, cdiv(qb * 128 + 128, 64)) for qb in range(nqb)]
        f.update(k=1, n=items / CUS, tot=B * H * sum(lens), longest=max(lens))
    else:
        k = split_parts(items, nqb, T)
        f.update(k=k, n=items * k / CUS, tot=items * T, longest=T / k)
    return f


def cbal_applies(r):  # fa_fwd_16.hip fwd_16_split_plan: balanced causal pairs (round 6), the plan's own gate
    if not r["causal"] or r["D"] not in (64, 128) or r["Skv"] < r["Sq"]:
        return False
    nqb = cdiv(r["Sq"], 128)
    items = r["B"] * r["H"] * nqb
    if nqb < 2:
        return False
    return (nqb >= 8 and items <= 4 * CUS) if r["D"] == 128 else (nqb >= 16 and items <= 2 * CUS)


def cbal_features(r):
    nqb, T = cdiv(r["Sq"], 128), cdiv(r["Skv"], 64)
    vbytes = r["B"] * r["H"] * r["Skv"] * r["D"] * 2
    return dict(h=0.5 * (min(T, 2) + min(T, cdiv((nqb - 1) * 128 + 128, 64))), n=r["B"] * r["H"] * nqb / CUS, fp16=r["fp16"],
                cast=1 if (not r["fp16"] and vbytes >= (16 << 20) and r["Sq"] >= 1024) else 0, vmb=vbytes / 1e6)


def cbal_predict(p, f, base):  # base: the (head_dim, causal) row of the 128-row table (cast pass, fp16 factor)
    a1, b1, a2, b2 = p
    full = int(f["n"] // 2)
    rest = f["n"] - 2 * full
    one, two = a1 + b1 * f["h"], a2 + b2 * f["h"]
    body = full * two + ((one if rest <= 1.0 else two) if rest > 1e-9 else 0.0)
    if f["fp16"]:
        body *= base[8]
    return f["cast"] * (base[5] + f["vmb"] * 2 / base[6]) + body


def r128_predict(p, f):
    t0, c_item, tau1, f2, f3, cast_a, cast_tbps, c_tail, g16 = p
    if f["fp16"]:
        tau1 = tau1 * g16  # fp16 operands: no conversion of V on its way into LDS
    R, n, fr = f["R"], f["n"], {0: 0.0, 1: 1.0, 2: f2, 3: f3}
    if f["causal"]:
        if n <= R:
            body = c_item + f["longest"] * tau1 * fr[max(1, math.ceil(n))]
        else:
            thr = f["tot"] / CUS * tau1 * fr[R] / R + n * c_item / R
            body = max(thr + c_tail * f["longest"] * tau1, c_item + f["longest"] * tau1 * fr[R])
    else:
        full = int(n // R)
        rest = math.ceil(n - full * R - 1e-9)
        L = f["longest"]
        body = full * (c_item + L * tau1 * fr[R]) + ((c_item + L * tau1 * fr[rest]) if rest > 0 else 0) + (c_tail * f["k"] if f["k"] > 1 else 0)
    return t0 + f["cast"] * (cast_a + f["vmb"] * 2 / cast_tbps) + body


def main():
    paths = sys.argv[1:] or sorted(glob.glob(str(ROOT / "profiles" / "r6" / "routing_random_*.jsonl")))
    recs = load(paths)
    print(f"{len(recs)} launches with both kernels measured, from {len(paths)} files")
    fit = {}
    for kind, feat, pred, x0, lo, hi in (
            ("w64", w64_features, w64_predict, [5, 4, 1.3, 8, 3, 4.0], [0, 0, 0.1, 0, 0, 0.5], [40, 30, 5, 60, 30, 20]),
            ("r128", r128_features, r128_predict, [5, 3, 1.0, 1.25, 1.8, 3, 4.0, 2.0, 0.9], [0, 0, 0.05, 1, 1, 0, 0.5, 0, 0.5], [40, 30, 5, 2, 3, 30, 20, 20, 1.2])):
        for D in (64, 128):
            for causal in (False, True):
                rs = [r for r in recs if r["D"] == D and r["causal"] == causal and not (kind == "r128" and cbal_applies(r))]
                F = [feat(r) for r in rs]
                y = np.log(np.array([r[kind] for r in rs]))
                sol = least_squares(lambda p: np.log(np.array([pred(p, f) for f in F])) - y, x0, bounds=(lo, hi))
                e = np.exp(np.abs(np.log(np.array([pred(sol.x, f) for f in F])) - y))
                print(f"{kind:4s} head_dim {D:3d} {'causal' if causal else 'full  '} n {len(rs):3d}: error median {np.median(e):.3f} p90 {np.quantile(e, 0.9):.3f} max {e.max():.3f}")
                fit[(kind, D, causal)] = sol.x
    for D in (64, 128):  # paired causal launches of the 128-row kernel: their own four constants per head dim
        rs = [r for r in recs if r["D"] == D and cbal_applies(r)]
        base = fit[("r128", D, True)]
        if len(rs) >= 6:
            F = [cbal_features(r) for r in rs]
            y = np.log(np.array([r["r128"] for r in rs]))
            sol = least_squares(lambda p: np.log(np.array([cbal_predict(p, f, base) for f in F])) - y, [10, 1.5, 20, 1.8], bounds=([0, 0.1, 0, 0.1], [60, 5, 80, 8]))
            e = np.exp(np.abs(np.log(np.array([cbal_predict(sol.x, f, base) for f in F])) - y))
            print(f"cbal head_dim {D:3d} n {len(rs):3d}: error median {np.median(e):.3f} p90 {np.quantile(e, 0.9):.3f} max {e.max():.3f}")
            fit[("cbal", D)] = sol.x
        else:
            fit[("cbal", D)] = np.array([8.3, 1.22, 11.7, 1.34] if D == 64 else [11.85, 1.64, 23.5, 1.97])
            print(f"cbal head_dim {D}: {len(rs)} launches -- the hand-fitted constants stay")
    loss = []
    for r in recs:
        pw = w64_predict(fit[("w64", r["D"], r["causal"])], w64_features(r))
        pr = (cbal_predict(fit[("cbal", r["D"])], cbal_features(r), fit[("r128", r["D"], True)]) if cbal_applies(r)
              else r128_predict(fit[("r128", r["D"], r["causal"])], r128_features(r)))
        loss.append((r["w64"] if pw < pr else r["r128"]) / min(r["w64"], r["r128"]))
    loss = np.array(loss)
    cur = np.array([r["dflt"] / min(r["w64"], r["r128"]) for r in recs])
    for name, a in (("routing by this model          ", loss), ("the dispatcher that was measured", cur)):
        print(f"{name}: chosen / best median {np.median(a):.3f} p90 {np.quantile(a, .9):.3f} p99 {np.quantile(a, .99):.3f} max {a.max():.3f}; more than 5 % behind: {(a > 1.05).mean() * 100:.1f} %")
    for kind, typ in (("w64", "W64Cost kW64Cost"), ("r128", "R128Cost kR128Cost")):
        rows = []
        for D in (64, 128):
            rows.append("    {" + ", ".join("{" + ", ".join(f"{x:.4f}f" for x in fit[(kind, D, c)]) + "}" for c in (False, True)) + "}")
        print(f"static const {typ}[2][2] = {{\n" + ",\n".join(rows) + "};")
    print("static const CbalCost kCbalCost[2] = {" + ", ".join("{" + ", ".join(f"{x:.4f}f" for x in fit[("cbal", D)]) + "}" for D in (64, 128)) + "};")


if __name__ == "__main__":
    main()
