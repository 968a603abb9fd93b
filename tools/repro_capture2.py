import ast, os, sys, faulthandler, tempfile
faulthandler.enable()
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests._util import load_golden, sub_state_dict, t
from graphnet_classifier_amd import GNN as G
from graphnet_classifier_amd import train as T
g = load_golden("g8_training_run.npz")
kw = ast.literal_eval(bytes(g["kwargs_json"]).decode())
m = G.CombinedModel(G.GraphNet(**kw), num_nodes=64, classes=2)
m.load_state_dict(sub_state_dict(g, "before/"), strict=True)
pos, ei = t(g["pos"]), t(g["edge_index"])
xs = [t(g["x0"]), t(g["x1"])]
ds = [((xs[k], pos, ei), torch.tensor(int(g["labels"][k]))) for k in range(2)]
mode = os.environ.get("MODE", "train")
if mode == "train":
    r = T.train(m, ds, 2, patience=5, output_path=tempfile.mkdtemp(), capture=True)
    print("train ok", r["captured"], r["avg_loss"], flush=True)
else:
    flat = T.FlatParameters(m); opt = T.FusedAdam(flat)
    crit = torch.nn.CrossEntropyLoss()
    loss_sum = torch.zeros((), dtype=torch.float64, device="cuda:0")
    if mode == "eager_side":
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            s, l = ds[0]
            s2 = (s[0].to("cuda:0"), s[1].to("cuda:0"), s[2])
            loss = crit(m(s2), l.to("cuda:0")); opt.zero_grad(); loss.backward(); opt.step(); loss_sum += loss.detach().double()
        torch.cuda.current_stream().wait_stream(side)
    if mode == "eager_nograd":
        s, l = ds[0]
        s2 = (s[0].to("cuda:0"), s[1].to("cuda:0"), s[2])
        with torch.no_grad():
            m(s2)
    if mode in ("eager_first", "eager_first_del"):
        s, l = ds[0]
        s2 = (s[0].to("cuda:0"), s[1].to("cuda:0"), s[2])
        loss = crit(m(s2), l.to("cuda:0")); opt.zero_grad(); loss.backward(); opt.step(); loss_sum += loss.detach().double()
        if mode == "eager_first_del":  # drop the old graph: its AccumulateGrad nodes (stream = legacy default) die with it
            del loss
    c = T.CapturedTrainStep(m, opt, crit, ds[1][0], ds[1][1], loss_sum)
    c(ds[0][0], ds[0][1]); torch.cuda.synchronize()
    print("captured ok", mode, float(loss_sum), flush=True)
