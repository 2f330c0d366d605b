import os, sys, torch
sys.path.insert(0, "audiodeepfake-detection_amd"); sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
from audiofakedetect.lcnn import LCNN
from oracle import torch_ref
from recipes import fill_state_dict
g = torch.load("tests/golden/dcnn_lcnn_eval.pt", map_location="cpu")
net = LCNN(); net.load_state_dict(fill_state_dict(g["shapes"]), strict=True)
ref = torch_ref.LCNNRef().double(); ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in net.state_dict().items()})
for m in (net, ref):
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
net.cuda()
x = g["x"]; labels = torch.arange(x.shape[0]) % 2
out = net(x.cuda()); loss = torch.nn.functional.cross_entropy(out, labels.cuda()); loss.backward()
o2 = ref(x.double()); l2 = torch.nn.functional.cross_entropy(o2, labels); l2.backward()
print("logit err", (out.detach().cpu().double() - o2.detach()).abs().max().item())
refp = dict(ref.named_parameters())
for k, p in net.named_parameters():
    d = (p.grad.cpu().double() - refp[k].grad); r = refp[k].grad
    print(f"{k:40s} relL2 {d.norm().item()/(r.norm().item()+1e-30):.3e}  max {d.abs().max().item():.3e} / {r.abs().max().item():.3e}")
