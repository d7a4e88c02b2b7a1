"""PMC / trace target: `python3 scratch/run_fwd.py B n [dtype]` runs n forwards of B synthetic images (tools/pmc_traffic.py uses the last)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blindshadowremoval_amd import Generator, init_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
gen = Generator(dtype=dtype).load_weights(init_weights(1))
torch.manual_seed(0)
inp = torch.rand(B, 256, 256, 3).cuda(); uv = torch.rand(B, 256, 256, 3).cuda()
for _ in range(n): gen(inp, uv)
torch.cuda.synchronize()
