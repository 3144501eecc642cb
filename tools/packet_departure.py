#!/usr/bin/env python3
"""How far is the scalar light-pdf rule of the oracle's default mode (and of the device) from the reference's
8-lane packet rule?  Renders the same passes of a scene at full resolution (packet composition depends on how many
rays of a pass hit each shape, i.e. on the resolution) with oracle mode 0 (rectangle / triangle light pdf = the
un-clipped plane density, always) and mode 2 (the reference's packets: 0 when no lane of the packet hits the light,
one light index per packet) and reports the PSNR between the two images as passes accumulate.  Both modes draw the
same random streams, so with one light the paths are the same and only their weights differ.

    python tools/packet_departure.py c4 1024 --threads 6 --out profiles/r02_packet_departure_c4.json

CPU only (the oracle is the thing compared); takes minutes to an hour.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing_amd"))
import oracle  # noqa: E402
from v4ray_hip import scenes  # noqa: E402


def psnr(a, b):
    fa, fb = np.clip(np.sqrt(np.maximum(a, 0)), 0, 1), np.clip(np.sqrt(np.maximum(b, 0)), 0, 1)
    mse = float(np.mean((fa.astype(np.float64) - fb) ** 2))
    return 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("spp", type=int)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--block", type=int, default=32)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--flags", type=int, default=0, help="scene flags for the mode-0 side (1 = bounded light pdf)")
    ap.add_argument("--mode", type=int, default=0, help="the oracle mode compared with mode 2: 0 = scalar rule (flags pick the pdf), 3 = packet-statistics rule")
    ap.add_argument("--out", default="")
    ap.add_argument("--save-images", default="", help="npz of both mean images, box-filtered down 8x")
    args = ap.parse_args()
    fw, fh, _ = scenes.FULL_SIZE[args.workload]
    w, h = args.width or fw, args.height or fh
    s0 = scenes.load(args.workload, w, h, pretransform_triangles=True, flags=args.flags)
    d0 = s0.flatten(s0.renderer_param, s0.camera_param)
    s2 = scenes.load(args.workload, w, h, pretransform_triangles=True)
    d2 = s2.flatten(s2.renderer_param, s2.camera_param)
    prim, _ = oracle.trace_primary(d0)
    tri = np.zeros(d0.contents.n_prims, dtype=bool)
    for i in range(d0.contents.n_prims):
        tri[i] = (d0.contents.prims[i].type & 0xFF) == 4
    on_mesh = np.where(prim != 0xFFFFFFFF, tri[np.minimum(prim, len(tri) - 1)], False)
    a0 = np.zeros((h, w, 3), dtype=np.float32)
    a2 = np.zeros((h, w, 3), dtype=np.float32)
    done, rows, t_start = 0, [], time.time()
    marks = {m for m in (16, 64, 256, 1024, 2048, 4096, 8192) if m <= args.spp} | {args.spp}
    while done < args.spp:
        n = min(args.block, args.spp - done)
        oracle.render(d0, n, first_pass=done, threads=args.threads, mode=args.mode, accum=a0)
        oracle.render(d2, n, first_pass=done, threads=args.threads, mode=2, accum=a2)
        done += n
        if done in marks:
            m0, m2 = a0 / done, a2 / done
            row = {"spp": done, "psnr_db": round(psnr(m0, m2), 2), "mean_mode0": float(m0.mean()), "mean_mode2": float(m2.mean()),
                   "mean_ratio_mode2_over_mode0": float(m2.mean() / m0.mean()),
                   "psnr_db_mesh_pixels": round(psnr(m0[on_mesh], m2[on_mesh]), 2) if on_mesh.any() else None,
                   "mean_ratio_mesh_pixels": float(m2[on_mesh].mean() / m0[on_mesh].mean()) if on_mesh.any() else None,
                   "psnr_db_other_pixels": round(psnr(m0[~on_mesh], m2[~on_mesh]), 2),
                   "mean_ratio_other_pixels": float(m2[~on_mesh].mean() / m0[~on_mesh].mean()),
                   "seconds": round(time.time() - t_start)}
            rows.append(row)
            print(json.dumps(row), flush=True)
            if args.out:
                with open(args.out, "w") as f:
                    json.dump({"workload": args.workload, "width": w, "height": h, "mode0_flags": args.flags, "compared_mode": args.mode,
                               "mesh_pixel_fraction": float(on_mesh.mean()),
                               "what": "PSNR (sqrt-gamma display values, clipped to [0,1]) between oracle mode `compared_mode` (keys say mode0; 0 = scalar rule, 3 = packet-statistics rule) and mode 2 "
                                       "(the reference's 8-lane packets), same passes, same random streams",
                               "rows": rows}, f, indent=1)
            if args.save_images:
                f8 = lambda a: a[:h // 8 * 8, :w // 8 * 8].reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3)).astype(np.float32)
                np.savez_compressed(args.save_images, mode0=f8(m0), mode2=f8(m2), spp=done)


if __name__ == "__main__":
    main()
