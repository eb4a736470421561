"""Developer aid: device time of the library's radix sort (gsr_sort_pairs_u32, the integer core of the binning) at the
shapes the frame presets produce, checked bit-exactly against torch's stable sort first.  GSR_LIB_PATH selects the
build, so two builds can be compared on one box:
    python3 scripts/sort_bench.py ; GSR_LIB_PATH=gaussmart_amd/lib/libgsr_hip_base.so python3 scripts/sort_bench.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussmart_amd.knn import sort_pairs_u32  # noqa: E402

SHAPES = [("depth 1M x 32b", 1_000_000, 32, "depth"), ("tile 3.2M x 13b", 3_200_000, 13, "tile"),
          ("depth 5M x 32b", 5_000_000, 32, "depth"), ("tile 22.6M x 12b", 22_600_000, 12, "tile"),
          ("depth 300k x 32b", 300_000, 32, "depth"), ("tile 3.6M x 13b", 3_600_000, 13, "tile")]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    out = {}
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, n, bits, kind in SHAPES:
        if only not in name:
            continue
        if kind == "depth":   # float bits of depths in [2, 10], as the synthetic scene has them
            keys = (torch.rand(n, generator=g) * 8 + 2).view(torch.int32).to(dev)
        else:
            keys = torch.randint(0, (1 << bits) - 30, (n,), generator=g, dtype=torch.int32).to(dev)
        ko, vo = sort_pairs_u32(keys, None, 0, bits)
        ref_k, ref_i = torch.sort(keys.to(torch.int64), stable=True)
        assert torch.equal(ko.to(torch.int64), ref_k) and torch.equal(vo.to(torch.int64), ref_i), name
        for _ in range(3):
            sort_pairs_u32(keys, None, 0, bits)
        torch.cuda.synchronize()
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            sort_pairs_u32(keys, None, 0, bits)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / reps
        passes = (bits + 7) // 8
        out[name] = {"us": round(us, 1), "passes": passes, "GBps_algorithmic": round(n * 8 * 2 * passes / us / 1e3, 1)}
        print(name, out[name], flush=True)
    print(json.dumps({"lib": os.environ.get("GSR_LIB_PATH", "in-tree"), "sort_us": out}))


if __name__ == "__main__":
    main()
