#!/usr/bin/env python3
"""Root-cause probe of the hipStreamEndCapture crash: plain PyTorch-ROCm, no kernel of this
repo.  Each MODE runs in its own process (the failing ones take the process down):

  stale_default   one eager step on the legacy default stream, `loss` KEPT ALIVE (its graph pins the parameters'
                  AccumulateGrad nodes, whose stream is the stream they were created on), then capture fwd+bwd
  fresh_default   the same eager step, but the old graph is dropped (`del loss`) before the capture
  stale_side      the eager step runs on a side stream, `loss` kept alive
  alias_default   stale_default, but the captured step differentiates private leaf aliases of the parameters
                  (torch.func.functional_call), so no node of the caller's history takes part
"""
import faulthandler
import os
import sys

faulthandler.enable()
import torch  # noqa: E402

mode = os.environ.get("MODE", "stale_default")
dev = "cuda:0"
torch.manual_seed(0)
model = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 2)).to(dev)
x = torch.randn(16, 32, device=dev)
y = torch.randint(0, 2, (16,), device=dev)
crit = torch.nn.CrossEntropyLoss()


def step(params=None):
    if params is None:
        out = model(x)
    else:
        out = torch.func.functional_call(model, params, (x,))
    loss = crit(out, y)
    loss.backward()
    return loss


if mode in ("stale_default", "fresh_default", "alias_default"):
    loss = step()
elif mode == "stale_side":
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        loss = step()
    torch.cuda.current_stream().wait_stream(s)
if mode == "fresh_default":
    del loss
torch.cuda.synchronize()
print("eager step done", mode, flush=True)

alias = None
if mode == "alias_default":
    alias = {k: p.detach().requires_grad_() for k, p in model.named_parameters()}
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step(alias)
torch.cuda.current_stream().wait_stream(side)
print("warm-up done", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step(alias)
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed ok", mode, flush=True)
