"""Developer aid: torch's autograd engine starts one thread per visible GPU at the first backward(), which opens the
device files even in a process that only does CPU work.  Which environment keeps a worker off the GPU?"""
import os, sys


def fds():
    out = []
    for f in os.listdir("/proc/self/fd"):
        try:
            t = os.readlink(f"/proc/self/fd/{f}")
        except OSError:
            continue
        if "kfd" in t or "dri" in t:
            out.append(t)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        import torch
        x = torch.randn(10, requires_grad=True)
        (x * x).sum().backward()
        print(sys.argv[2], {k: os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")},
              "-> after backward:", fds(), flush=True)
    else:
        import subprocess
        for name, env in (("plain", {}), ("hip_empty", {"HIP_VISIBLE_DEVICES": ""}), ("cuda_empty", {"CUDA_VISIBLE_DEVICES": ""}),
                          ("rocr_empty", {"ROCR_VISIBLE_DEVICES": ""}), ("hip_-1", {"HIP_VISIBLE_DEVICES": "-1"}),
                          ("all_empty", {"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})):
            subprocess.run([sys.executable, __file__, "child", name], env={**os.environ, **env})
