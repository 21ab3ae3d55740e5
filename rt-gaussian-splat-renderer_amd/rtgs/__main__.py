"""Headless driver with the reference CLI's flags (reference: src/rtgs/__main__.py:31-258).

The reference opens a ``ti.GUI`` window and performs one depth peel per GUI frame until
``--sample`` samples are accumulated (__main__.py:253-257).  There is no display on an MI355X
node, so this driver runs the same loop without the window, for one orbit-camera pose given on
the command line (the viewer's mouse state: theta, phi, radius, cursor), and writes the display
buffer as a PNG or PPM.  ``--synthetic N`` renders a seeded random scene instead of a PLY.

    python -m rtgs -o scene.ply -r 960,540 -f 90 -d 16 --out frame.png
"""

from __future__ import annotations

import argparse
import logging
import os
import pathlib
import struct
import time
import zlib

import numpy as np

logger = logging.getLogger("rtgs")


def orbit_pose(theta: float, phi: float, r: float, cursor=(0.0, 0.0, 0.0)):
    """Orbit camera of the reference viewer (__main__.py:120-142): position on a sphere of
    radius r about `cursor`, looking at it, world Z up; returns (position, quaternion xyzw)."""
    pos = np.array([r * np.cos(theta) * np.sin(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(phi)])
    look = -pos / np.linalg.norm(pos)
    right = np.array([-np.sin(theta), np.cos(theta), 0.0])
    up = np.cross(right, look)
    m = np.array([right, up, -look]).T      # columns: camera +x, +y, +z in world space
    # rotation matrix -> unit quaternion (Shepperd), scalar last
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
    else:
        i = int(np.argmax(np.diag(m)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + m[i, i] - m[j, j] - m[k, k]) * 2
        q = [0.0, 0.0, 0.0, (m[k, j] - m[j, k]) / s]
        q[i], q[j], q[k] = 0.25 * s, (m[j, i] + m[i, j]) / s, (m[k, i] + m[i, k]) / s
    return pos + np.asarray(cursor, dtype=np.float64), np.asarray(q, dtype=np.float64)


def write_image(path: str, disp) -> None:
    """disp: (W, H, 3) float, [i, j] with (0, 0) bottom-left -> row-major top-down 8-bit file."""
    img = np.clip(np.asarray(disp, dtype=np.float32), 0.0, 1.0)
    img = (np.transpose(img, (1, 0, 2))[::-1] * 255.0 + 0.5).astype(np.uint8)   # img[H-1-j, i]
    h, w, _ = img.shape
    if path.lower().endswith(".ppm"):
        with open(path, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (w, h) + img.tobytes())
        return
    raw = b"".join(b"\x00" + img[r].tobytes() for r in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main(argv=None) -> int:
    logging.basicConfig(level=os.environ.get("LOG_LEVEL", "WARNING"))
    ap = argparse.ArgumentParser("rtgs", description="HIP/MI355X implementation of 3D Gaussian ray tracing.")
    ap.add_argument("-o", "--open", type=pathlib.Path, help="Path to the .ply Gaussian splatting scene file.")
    ap.add_argument("-r", "--res", type=lambda s: tuple(map(int, s.split(","))), default=(960, 540),
                    help="Render resolution")
    ap.add_argument("-f", "--fov", type=float, default=90, help="Vertical FOV in degree.")
    ap.add_argument("-s", "--sample", type=int, default=1, help="Render sample rate.")
    ap.add_argument("-d", "--depth", type=int, default=16, help="Render sample depth.")
    ap.add_argument("-v", "--bvh", type=int, default=1024, help="BVH size (accepted, unused)")
    ap.add_argument("--scale", type=float, default=1, help="Global Gaussian Scale")
    ap.add_argument("--theta", type=float, default=0.0, help="orbit azimuth (viewer default 0)")
    ap.add_argument("--phi", type=float, default=np.pi / 2, help="orbit polar angle (viewer default pi/2)")
    ap.add_argument("--radius", type=float, default=1.0, help="orbit radius (viewer default 1)")
    ap.add_argument("--cursor", type=lambda s: tuple(map(float, s.split(","))), default=(0.0, 0.0, 0.0))
    ap.add_argument("--synthetic", type=int, default=0, help="render N seeded random Gaussians instead of a file")
    ap.add_argument("--out", default="rtgs.png", help="output image (.png or .ppm)")
    ap.add_argument("--one-pass", action="store_true", help="whole samples in one launch (sample_all)")
    args = ap.parse_args(argv)

    from . import Camera, RayTracer, Scene, synthetic

    res = tuple(args.res)
    focal = (res[1] / 2) / np.tan(args.fov * np.pi / 360)          # __main__.py:91-92
    scene = Scene(args.bvh, 4, 16)                                  # __main__.py:98
    if args.synthetic:
        scene.set_arrays(**synthetic.make_scene(args.synthetic, res[0] / res[1], 1))
    elif args.open:
        scene.load_file(args.open, args.scale)
    else:
        ap.error("give a scene with -o/--open or --synthetic N")
    pos, quat = orbit_pose(args.theta, args.phi, args.radius, args.cursor)
    camera = Camera(pos, quat, res, (focal, focal))                 # __main__.py:110-115
    tracer = RayTracer(res, scene, camera)                          # __main__.py:147
    import torch
    t0 = time.perf_counter()
    while tracer.num_samples < args.sample:                         # __main__.py:253-256
        if args.one_pass:
            tracer.sample_all(args.depth)
        else:
            tracer.sample(args.depth)
    tracer.generate_disp_buffer(tracer.num_samples, tracer.num_steps, args.depth)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    write_image(args.out, tracer.disp_buf.cpu().numpy())
    print(f"{scene.num_gaussians} Gaussians, {res[0]}x{res[1]}, {args.sample} sample(s) x depth {args.depth}: "
          f"{dt * 1e3:.2f} ms -> {args.out}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
