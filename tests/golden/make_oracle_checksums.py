"""Writes tests/golden/oracle_grad_checksums.json: per oracle-farm case (tests/oracle_farm.py) and gradient tensor the pair
(sum, sum of absolute values) of the fp64 oracle gradients.  The GPU parity tests run the oracle LIVE for every case and
check its result against these sums (oracle_farm.check_against_committed_checksums): a change of the oracle, of a scene
builder, or of the CPU kernels underneath that moves the reference gradients is caught as such.

    python tests/golden/make_oracle_checksums.py [workers]        (a few minutes on 8 cores; no GPU, no reference import)

The oracle is the build's own restatement (oracle/surfel_ref.py, "parity unpinned" against upstream: SURVEY 8(c)); these
sums pin the oracle against ITSELF over time, nothing more.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("FARM_WORKERS", sys.argv[1] if len(sys.argv) > 1 else "6")
    import oracle_farm as F
    # importing the test modules registers their cases
    import test_gpu_rasterizer, test_gpu_deep_lists, test_gpu_wide_payload  # noqa: F401
    # the checksums only need the fp64 pass
    for sp in F.FARM.specs.values():
        sp["want32"], sp["sens_tols"] = False, ()
    F.FARM.start()
    out = {}
    for key in sorted(F.FARM.specs):
        res = F.FARM.get(key)
        out[key] = res["checksum"]
        print(key, {k: f"{v[1]:.6e}" for k, v in res["checksum"].items()}, flush=True)
    F.FARM.shutdown()
    with open(os.path.join(HERE, "oracle_grad_checksums.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"{len(out)} cases written")


if __name__ == "__main__":
    main()
