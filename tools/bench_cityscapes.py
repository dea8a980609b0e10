"""BASELINE.json configs[3] shape on ONE GPU: Cityscapes-like 19-class RGB 512x1024, 2+2 images per step
(16 images over 8 GPUs).  Functional / timing check of the same step at that scale."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, random, numpy as np
from arco_amd import train_arco_2d as T
random.seed(1337); np.random.seed(1337); torch.manual_seed(1337)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
args = T.build_parser().parse_args(["--batch_size", str(b), "--queue_size", "4096", "--synthetic", "1", "--num_classes", "19",
                                    "--in_chns", "3"])
args.patch_size = [512, 1024]
st = T.ArcoStep2D(args, "cuda:0")
l, ll = T.synthetic_batch(b, args.patch_size, 19, 1, "cuda:0", in_chns=3)
u, _ = T.synthetic_batch(b, args.patch_size, 19, 2, "cuda:0", in_chns=3)
for _ in range(4): st.step(l, ll, u)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): loss, reco = st.step(l, ll, u)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
assert bool(torch.isfinite(loss))
print(f"Cityscapes-shaped step b={b} ({2*b} images 512x1024, 19 classes): {dt*1e3:.1f} ms/step  loss {float(loss):.4f} reco {float(reco):.4f}  "
      f"mem {torch.cuda.max_memory_allocated()/1e9:.1f} GB  terms {({k: round(float(v), 4) for k, v in st.last_terms.items()})}")
