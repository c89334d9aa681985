"""Per-frame latency of causal streaming inference (gast_hip/streaming.py) next to the reference's way of producing one pose per
frame (a window forward over the last receptive_field frames, gen_skes.py:43-69 / tools/inference.py:73-91) on the same model.
Usage (GPU box): python scripts/stream_bench.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gast-net-3dposeestimation_amd')):
    sys.path.insert(0, p)
import torch
from bench import adj_from_parents, PARENTS17
from model.gast_net import SpatioTemporalModelOptimized1f
from gast_hip.streaming import CausalStream

kl, kr = [4, 5, 6, 11, 12, 13], [1, 2, 3, 14, 15, 16]
out = []
for arc, ch in (([3, 3, 3], 128), ([3, 3, 3, 3], 64)):
    torch.manual_seed(0)
    m = SpatioTemporalModelOptimized1f(adj_from_parents(PARENTS17), 17, 2, 17, filter_widths=arc, causal=True, channels=ch, dropout=0.25).cuda().eval()
    rf = m.receptive_field()
    g = torch.Generator().manual_seed(1)
    clip = (torch.rand(1, 300, 17, 2, generator=g) * 2 - 1).cuda()
    st = CausalStream(m, batch=1, flip=(kl, kr, kl, kr))
    st.run(clip[:, :40])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(40, 300):
        st.push(clip[:, t])
    torch.cuda.synchronize(); stream_ms = (time.perf_counter() - t0) / 260 * 1e3
    # the reference's way: window of rf frames (+ mirrored copy) through the single-frame model, per frame
    win = torch.cat([clip[:, :rf], clip[:, :rf]], 0).contiguous()
    with torch.no_grad():
        for _ in range(5): m(win)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): m(win)
        torch.cuda.synchronize(); window_ms = (time.perf_counter() - t0) / 50 * 1e3
    out.append(dict(arc=arc, channels=ch, receptive_field=rf, stream_ms_per_frame=round(stream_ms, 4), window_forward_ms_per_frame=round(window_ms, 4),
                    note='batch 1 + mirrored copy (flip TTA), eval mode, dtype %s; stream = hipGraph replay of the one-frame step' % os.environ.get('GAST_HIP_DTYPE', 'fp32')))
print(json.dumps(out))
