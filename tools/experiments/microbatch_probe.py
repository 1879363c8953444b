"""GPU box: one optimiser step over 64 graphs as ONE minibatch vs as TWO 32-graph micro-batches whose forward / backward chains run
concurrently on two streams (gradients accumulate into the same flat buffer, each half's loss weighted 1/2).  Is the chip better
filled by two dependent chains than by one?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
import torch
from buglab.data.collate import collate_samples, to_device
from buglab.data.synthetic import make_samples
from buglab.models import hip_ops
from buglab.models.gnn import build_gnn_mlp_module
from buglab.runtime.optim import FlatAdam

dev = torch.device("cuda")
H, B = int(os.environ.get("HID", 128)), int(os.environ.get("GRAPHS", 64))
samples = make_samples(B, seed=1000, num_nodes=2000, num_messages=10000, num_edge_types=16)
mb_full = to_device(collate_samples(samples, 16), dev)
halves = [to_device(collate_samples(samples[: B // 2], 16), dev), to_device(collate_samples(samples[B // 2:], 16), dev)]
module = build_gnn_mlp_module(H, 8, 16, dropout_rate=0.2, dropout_base_seed=0, embedder_dropout_rate=0.0).to(dev).train()
opt = FlatAdam(module.parameters())
s_main = hip_ops.use_step_stream(dev) or torch.cuda.current_stream()
streams = [torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=-1)]


def step_one():
    opt.zero_grad()
    loss = module(**mb_full)
    loss.backward()
    opt.step()
    return loss


def step_two(interleave: bool):
    opt.zero_grad()
    cur = torch.cuda.current_stream()
    losses = []
    for st in streams:
        st.wait_stream(cur)
    if interleave:  # forward A, forward B, backward A, backward B
        for st, mb in zip(streams, halves):
            with torch.cuda.stream(st):
                losses.append(module(**mb) * 0.5)
        for st, l in zip(streams, losses):
            with torch.cuda.stream(st):
                l.backward()
    else:  # A entirely, then B entirely (B's forward overlaps A's backward)
        for st, mb in zip(streams, halves):
            with torch.cuda.stream(st):
                l = module(**mb) * 0.5
                l.backward()
                losses.append(l)
    for st in streams:
        cur.wait_stream(st)
    opt.step()
    return losses[0] + losses[1]


def timed(f, n=20, warm=5):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        l = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(l.detach())


for rnd in range(2):
    ms, l = timed(step_one)
    print(f"one minibatch of {B}:            {ms:7.3f} ms/step  {B / ms * 1e3:8.1f} graphs/s  loss {l:.4f}")
    for inter in (True, False):
        ms, l = timed(lambda: step_two(inter))
        print(f"two micro-batches ({'fwd A, fwd B, bwd A, bwd B' if inter else 'A then B'}): {ms:7.3f} ms/step  {B / ms * 1e3:8.1f} graphs/s  loss {l:.4f}")
