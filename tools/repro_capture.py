import ast, os, sys, faulthandler
faulthandler.enable()
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests._util import load_golden, sub_state_dict, t
from graphnet_classifier_amd import GNN as G
from graphnet_classifier_amd.train import FlatParameters, FusedAdam
from graphnet_classifier_amd.topology import get_topology
g = load_golden("g8_training_run.npz")
kw = ast.literal_eval(bytes(g["kwargs_json"]).decode())
width = int(os.environ.get("W", "0"))
if width:
    for k in list(kw):
        if "dim" in k: kw[k] = width
m = G.CombinedModel(G.GraphNet(**kw), num_nodes=64, classes=2)
if not width:
    m.load_state_dict(sub_state_dict(g, "before/"), strict=True)
dev = "cuda:0"
x, pos, ei = t(g["x0"]).to(dev), t(g["pos"]).to(dev), t(g["edge_index"]).to(dev)
label = torch.tensor(0, device=dev)
mode = os.environ.get("MODE", "full")
flat = FlatParameters(m); opt = FusedAdam(flat)
crit = torch.nn.CrossEntropyLoss()
topo = get_topology(ei, 64, dev)
def one_step():
    logits = m((x, pos, ei))
    if mode == "fwd": return logits
    loss = crit(logits, label)
    opt.zero_grad()
    loss.backward()
    if mode == "fwdbwd": return logits
    opt.step()
    return logits
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): one_step()
torch.cuda.current_stream().wait_stream(side)
opt.zero_grad()
print("warm-up done", flush=True)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    one_step()
print("captured", flush=True)
graph.replay(); torch.cuda.synchronize()
print("replayed ok", mode, width, flush=True)
