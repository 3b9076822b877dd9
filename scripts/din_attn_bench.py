"""Per-launch time of the fused DIN attention passes (fx_din_attn.hip) at BASELINE configs[3] shape.
usage: [FX_DIN_ATTN_WAVES=n] [FX_DIN_ATTN_BWD_WAVES=n] python scripts/din_attn_bench.py [B L E H]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fuxictr_amd import ops  # noqa: E402

B, L, E, H = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else (4096, 50, 16, 64)
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
q = torch.randn(B, E, generator=g).to(dev)
K = torch.randn(B, L, E, generator=g).to(dev)
W1 = (torch.randn(H, 4 * E, generator=g) * 0.3).to(dev)
b1 = torch.randn(H, generator=g).to(dev)
alpha = (torch.rand(H, generator=g) - 0.5).to(dev)
W2 = torch.randn(H, generator=g).to(dev)
b2 = torch.randn(1, generator=g).to(dev)
mask = (torch.rand(B, L, generator=g) > 0.3).to(torch.int32).to(dev)
rm, rv = torch.zeros(H, device=dev), torch.ones(H, device=dev)
ws = torch.empty(ops.din_attn_workspace_floats(B, L, E, H), device=dev)
sums = torch.empty(2 * H + 1, device=dev)
stats = torch.empty(2 * H, device=dev)
a = torch.empty(B, L, device=dev)
out = torch.empty(B, E, device=dev)
da = torch.randn(B, L, generator=g).to(dev)
dKp = torch.randn(B, L, E, generator=g).to(dev)
sums5 = torch.empty(5 * H, device=dev)
dq = torch.empty(B, E, device=dev)
dK = torch.empty(B, L, E, device=dev)
dW = torch.empty(H * 4 * E + H, device=dev)
dout = torch.randn(B, E, generator=g).to(dev)

steps = [
    ("stats (2 launches)", lambda: ops.din_attn_stats(q, K, W1, b1, sums, ws)),
    ("dice_stats_from_sums", lambda: ops.dice_stats_from_sums(sums, H, B * L, 0.01, True, rm, rv, stats)),
    ("fwd apply + pooling", lambda: ops.din_attn_fwd(q, K, W1, b1, alpha, 1e-9, stats, W2, b2, mask, a, out)),
    ("bwd sums (2 launches)", lambda: ops.din_attn_bwd_sums(q, K, W1, b1, alpha, 1e-9, stats, W2, mask, dout, da,
                                                            sums5, ws)),
    ("bwd apply (2 launches)", lambda: ops.din_attn_bwd(q, K, W1, b1, alpha, 1e-9, True, stats, W2, mask, a, dout,
                                                        da, sums5, B * L, dq, dK, dW, ws)),
]
for _, f in steps:
    f()
torch.cuda.synchronize()
total = 0.0
print("B=%d L=%d E=%d H=%d  waves=%s bwd_waves=%s" % (B, L, E, H, os.environ.get("FX_DIN_ATTN_WAVES", "dflt"),
                                                       os.environ.get("FX_DIN_ATTN_BWD_WAVES", "dflt")) + " fwd_waves=%s" % os.environ.get("FX_DIN_ATTN_FWD_WAVES", "dflt"))
for name, f in steps:
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / 20)
    total += best
    print("  %-26s %8.1f us" % (name, best))
print("  %-26s %8.1f us" % ("sum", total))
