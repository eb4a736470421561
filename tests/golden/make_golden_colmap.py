#!/usr/bin/env python3
"""Writes a small synthetic COLMAP sparse model (binary and text) and records what the REFERENCE's
own loader (scene/colmap_loader.py:83-242) reads from it.  Run once in the build container; the
model files and the .npz travel, the reference does not."""
import os
import struct
import sys

import numpy as np

REF = os.environ.get("GSR_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
OUT = os.path.dirname(os.path.abspath(__file__))


def write_model(root):
    rng = np.random.default_rng(7)
    b = os.path.join(root, "bin", "sparse", "0"); t = os.path.join(root, "txt", "sparse", "0")
    os.makedirs(b, exist_ok=True); os.makedirs(t, exist_ok=True)
    cams = [(1, 1, 640, 480, [500.0, 510.0, 320.0, 240.0]), (2, 0, 800, 600, [700.0, 400.0, 300.0])]
    with open(os.path.join(b, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(cams)))
        for cid, mid, w, h, par in cams:
            f.write(struct.pack("<iiQQ", cid, mid, w, h)); f.write(struct.pack("<%dd" % len(par), *par))
    with open(os.path.join(t, "cameras.txt"), "w") as f:
        f.write("# Camera list\n")
        for cid, mid, w, h, par in cams[:1]:
            f.write(f"{cid} PINHOLE {w} {h} " + " ".join(repr(p) for p in par) + "\n")
    imgs = []
    for i in range(5):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        tv = rng.normal(size=3) * 3
        n2d = int(rng.integers(0, 6))
        xys = rng.uniform(0, 600, size=(n2d, 2)); ids = rng.integers(-1, 50, size=n2d)
        imgs.append((i + 1, q, tv, 1 if i % 2 == 0 else 2, f"img_{i:03d}.png", xys, ids))
    with open(os.path.join(b, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(imgs)))
        for iid, q, tv, cid, name, xys, ids in imgs:
            f.write(struct.pack("<idddddddi", iid, *q, *tv, cid)); f.write(name.encode() + b"\x00")
            f.write(struct.pack("<Q", len(ids)))
            for (x, y), pid in zip(xys, ids):
                f.write(struct.pack("<ddq", x, y, int(pid)))
    with open(os.path.join(t, "images.txt"), "w") as f:
        f.write("# Image list with two lines of data per image\n")
        for iid, q, tv, cid, name, xys, ids in imgs:
            f.write(f"{iid} " + " ".join(repr(float(v)) for v in q) + " " + " ".join(repr(float(v)) for v in tv) + f" 1 {name}\n")
            f.write(" ".join(f"{x!r} {y!r} {int(p)}" for (x, y), p in zip(xys.tolist(), ids)) + "\n")
    pts = []
    for i in range(40):
        xyz = rng.normal(size=3) * 2; rgb = rng.integers(0, 256, size=3); err = float(rng.uniform(0, 2))
        track = [(int(rng.integers(1, 6)), int(rng.integers(0, 5))) for _ in range(int(rng.integers(0, 4)))]
        pts.append((i + 1, xyz, rgb, err, track))
    with open(os.path.join(b, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(pts)))
        for pid, xyz, rgb, err, track in pts:
            f.write(struct.pack("<QdddBBBd", pid, *xyz, *[int(c) for c in rgb], err)); f.write(struct.pack("<Q", len(track)))
            for a, c in track:
                f.write(struct.pack("<ii", a, c))
    with open(os.path.join(t, "points3D.txt"), "w") as f:
        f.write("# 3D point list\n")
        for pid, xyz, rgb, err, track in pts:
            f.write(f"{pid} " + " ".join(repr(float(v)) for v in xyz) + " " + " ".join(str(int(c)) for c in rgb) + f" {err!r} " +
                    " ".join(f"{a} {c}" for a, c in track) + "\n")
    return b, t


def main():
    # the reference's scene/__init__.py pulls in plyfile (absent here); colmap_loader.py itself only
    # needs numpy, so the module file is loaded on its own
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_colmap_loader", os.path.join(REF, "scene", "colmap_loader.py"))
    cl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cl)
    read_intrinsics_binary, read_extrinsics_binary, read_points3D_binary = cl.read_intrinsics_binary, cl.read_extrinsics_binary, cl.read_points3D_binary
    read_intrinsics_text, read_extrinsics_text, read_points3D_text, qvec2rotmat = cl.read_intrinsics_text, cl.read_extrinsics_text, cl.read_points3D_text, cl.qvec2rotmat
    root = os.path.join(OUT, "colmap_small")
    b, t = write_model(root)
    out = {}
    for tag, d, ri, re, rp in (("bin", b, read_intrinsics_binary, read_extrinsics_binary, read_points3D_binary),
                               ("txt", t, read_intrinsics_text, read_extrinsics_text, read_points3D_text)):
        ext = "bin" if tag == "bin" else "txt"
        cams = ri(os.path.join(d, f"cameras.{ext}")); imgs = re(os.path.join(d, f"images.{ext}"))
        xyz, rgb, err = rp(os.path.join(d, f"points3D.{ext}"))
        out[f"{tag}_cam_ids"] = np.array(sorted(cams))
        for k in sorted(cams):
            out[f"{tag}_cam{k}_wh"] = np.array([cams[k].width, cams[k].height]); out[f"{tag}_cam{k}_params"] = cams[k].params
            out[f"{tag}_cam{k}_model"] = np.array(cams[k].model)
        out[f"{tag}_img_ids"] = np.array(sorted(imgs))
        for k in sorted(imgs):
            im = imgs[k]
            out[f"{tag}_img{k}_qvec"] = im.qvec; out[f"{tag}_img{k}_tvec"] = im.tvec; out[f"{tag}_img{k}_cam"] = np.array(im.camera_id)
            out[f"{tag}_img{k}_name"] = np.array(im.name); out[f"{tag}_img{k}_xys"] = im.xys.reshape(-1, 2); out[f"{tag}_img{k}_p3d"] = im.point3D_ids
            out[f"{tag}_img{k}_R"] = qvec2rotmat(im.qvec)
        out[f"{tag}_xyz"], out[f"{tag}_rgb"], out[f"{tag}_err"] = xyz, rgb, err
    np.savez(os.path.join(OUT, "colmap_small.npz"), **out)
    print("written", root)


if __name__ == "__main__":
    main()
