"""Developer aid: a few forward+backward passes of the fused photometric loss at the headline resolution (for rocprofv3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussmart_amd.fused_loss import photometric_loss
W, H, reps = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080)), int(os.environ.get("REPS", 5))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(3, H, W, device=dev, generator=g).requires_grad_(True)
y = torch.rand(3, H, W, device=dev, generator=g)
for _ in range(reps):
    loss, _, _ = photometric_loss(x, y, 0.2)
    loss.backward()
torch.cuda.synchronize()
print("done", float(loss))
