"""Generates the golden DCNN/LCNN vectors in tests/golden/ by importing the reference.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
(`python tests/golden/make_golden.py level14` writes only the level-14 file dcnn_level14_eval.pt).

The reference's ``models.py`` imports torchsummary / timm / torchaudio at module scope for
code paths that are out of scope here (check_dimensions, ASTModel, augmentations); those
three are replaced by empty stub modules so that ``DCNN`` / ``LCNN`` -- plain torch.nn
code -- import and run on CPU.  Only tensors are written out (inputs, weights as
state_dict tensors, outputs); no reference source or pickled reference classes.
"""

import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference_models():
    _stub("torchsummary", summary=lambda *a, **k: None)
    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.layers", to_2tuple=lambda x: (x, x), trunc_normal_=lambda *a, **k: None)
    ta = _stub("torchaudio")
    ta.functional = _stub("torchaudio.functional")
    ta.transforms = _stub("torchaudio.transforms")
    sys.path.insert(0, REF)
    # data_loader/utils import torchaudio only; load models + DotDict
    import importlib

    models = importlib.import_module("audiofakedetect.models")
    utils = importlib.import_module("audiofakedetect.utils")
    return models, utils.DotDict


def strip_prefix(sd):
    return {k.replace("module.", ""): v for k, v in sd.items()}


def dcnn_args(DotDict, input_dim, time_dim_add=0, flattend_size=320):
    return DotDict(
        input_dim=list(input_dim), ochannels1=64, ochannels2=64, ochannels3=96,
        ochannels4=128, ochannels5=32, kernel1=3, dropout_cnn=0.6, dropout_lstm=0.2,
        time_dim_add=time_dim_add, flattend_size=flattend_size, ddp=False,
    )


def record_routing(net):
    """Hooks that record, for one forward of the reference DCNN, every piecewise-linear routing
    decision the backward pass depends on, in forward order:

    ("pool", uint8 [B,C,Hp,Wp])  PReLU followed by MaxPool2d(2, 2): bits 0-1 = position dy*2+dx of
                                 the window's maximum, bit 2 = the winning pre-activation was <= 0;
    ("prelu", packed bits)       any other PReLU: pre-activation <= 0, numpy.packbits of the
                                 flattened mask, with the shape next to it.
    Only these decisions are stored (no activations)."""
    import torch.nn as nn

    records = []
    mods = []
    for seq_name in ("cnn", "dil_conv"):
        seq = getattr(net, seq_name)
        for i, m in enumerate(seq):
            if isinstance(m, nn.PReLU):
                nxt = seq[i + 1] if i + 1 < len(seq) else None
                mods.append((f"{seq_name}.{i}", m, isinstance(nxt, nn.MaxPool2d)))

    def make_hook(name, pooled):
        def hook(mod, inp, out):
            z = inp[0].detach()
            if pooled:
                _, flat = torch.nn.functional.max_pool2d(out.detach(), 2, 2, return_indices=True)
                w = z.shape[-1]
                row, col = flat // w, flat % w
                pos = (row % 2) * 2 + (col % 2)
                zwin = z.flatten(2).gather(2, flat.flatten(2)).reshape(flat.shape)
                code = (pos + 4 * (zwin <= 0)).to(torch.uint8)
                records.append({"name": name, "kind": "pool", "code": code})
            else:
                mask = (z <= 0).flatten().numpy()
                records.append({"name": name, "kind": "prelu", "shape": tuple(z.shape),
                                "bits": torch.from_numpy(np.packbits(mask))})
        return hook

    for name, m, pooled in mods:
        m.register_forward_hook(make_hook(name, pooled))
    return records


def main():
    models, DotDict = import_reference_models()
    torch.manual_seed(0)
    gold = {}

    ckpt_dir = "/root/reference/models"
    files = {f.split("_")[1]: os.path.join(ckpt_dir, f) for f in os.listdir(ckpt_dir)}

    # --- eval-mode logits with the shipped weights (stft: T=101, sym5: T=95 add=1) ---
    for tag, t_len, add in (("stft", 101, 0), ("packetssym5", 95, 1)):
        sd = strip_prefix(torch.load(files[tag], map_location="cpu")["MODEL_STATE"])
        net = models.DCNN(dcnn_args(DotDict, (4, 1, 256, t_len), time_dim_add=add))
        net.load_state_dict(sd, strict=True)
        net.eval()
        g = torch.Generator().manual_seed(100 + t_len)
        x = torch.randn(4, 1, 256, t_len, generator=g)
        with torch.no_grad():
            y = net(x)
        gold[f"shipped_{tag}"] = {
            "state_dict": {k: v.clone() for k, v in sd.items()},
            "x": x, "logits": y, "labels": y.argmax(-1),
            "time_dim_add": add,
        }

    # --- seeded random weights: eval logits for the three level-8 shapes ---
    for tag, t_len, add in (("stft", 101, 0), ("sym5", 95, 1), ("coif4", 109, 0)):
        torch.manual_seed(7 + t_len)
        net = models.DCNN(dcnn_args(DotDict, (4, 1, 256, t_len), time_dim_add=add))
        # non-trivial running stats
        net.train()
        with torch.no_grad():
            for _ in range(2):
                net(torch.randn(4, 1, 256, t_len))
        net.eval()
        x = torch.randn(4, 1, 256, t_len)
        with torch.no_grad():
            y = net(x)
        gold[f"random_{tag}"] = {
            "state_dict": {k: v.clone() for k, v in net.state_dict().items()},
            "x": x, "logits": y, "labels": y.argmax(-1), "time_dim_add": add,
        }

    # --- one train step, dropout p=0, batch 8: loss, grads, Adam-updated params ---
    torch.manual_seed(11)
    args = dcnn_args(DotDict, (8, 1, 256, 101))
    args.dropout_cnn = 0.0
    args.dropout_lstm = 0.0
    net = models.DCNN(args)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.randn(8, 1, 256, 101)
    labels = torch.randint(0, 2, (8,))
    opt = torch.optim.Adam(net.parameters(), lr=4e-4, weight_decay=1e-3)
    net.train()
    opt.zero_grad()
    routing = record_routing(net)
    out = net(x)
    loss = torch.nn.CrossEntropyLoss()(out, labels)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    opt.step()
    # the reference run's routing decisions, in forward order (see record_routing)
    torch.save({"routing": routing}, os.path.join(OUT, "dcnn_train_step_routing.pt"))
    gold["train_step"] = {
        "state_dict": sd0, "x": x, "labels": labels, "logits": out.detach(),
        "loss": loss.detach(), "grads": grads,
        "state_dict_after": {k: v.clone() for k, v in net.state_dict().items()},
        "lr": 4e-4, "weight_decay": 1e-3,
    }

    # --- LCNN eval logits ---
    sys.path.insert(0, OUT)
    from recipes import fill_state_dict

    torch.manual_seed(13)
    lcnn = models.LCNN(classes=2, in_channels=1, lstm_channels=256)
    shapes = {k: tuple(v.shape) for k, v in lcnn.state_dict().items()}
    lcnn.load_state_dict(fill_state_dict(shapes), strict=True)
    lcnn.eval()
    x = torch.randn(4, 1, 256, 101)
    with torch.no_grad():
        y = lcnn(x)
    gold["lcnn_eval"] = {"shapes": shapes, "x": x, "logits": y, "labels": y.argmax(-1)}

    for name, blob in gold.items():
        torch.save(blob, os.path.join(OUT, f"dcnn_{name}.pt"))
        print(name, {k: (tuple(v.shape) if torch.is_tensor(v) else type(v).__name__)
                     for k, v in blob.items()})


def level14():
    """Reference-class fixtures at the headline geometry (BASELINE configs[1] / [2]): the reference's DCNN
    (models.py:240-313) in eval mode on packet images of level 14 -- coif4 [2,1,16384,24] (time_dim 3) and sym5
    [2,1,16384,10] (time_dim 1), flattend_size 80 960.  Weights come from recipes.fill_dcnn_state_dict (rebuilt from names
    and shapes by the test), inputs from a seeded generator (rebuilt by the test, checked against the stored digest):
    the file holds shapes, seeds, digests, logits and labels."""
    models, DotDict = import_reference_models()
    sys.path.insert(0, OUT)
    from recipes import fill_dcnn_state_dict, level14_input

    gold = {}
    for tag, t_len in (("coif4", 24), ("sym5", 10)):
        net = models.DCNN(dcnn_args(DotDict, (2, 1, 16384, t_len), time_dim_add=0, flattend_size=80960))
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        sd = fill_dcnn_state_dict(shapes)
        net.load_state_dict(sd, strict=True)
        net.eval()
        x = level14_input(t_len)
        with torch.no_grad():
            y = net(x)
        gold[tag] = {"shapes": shapes, "t_len": t_len,
                     "x_digest": torch.stack([x.double().sum(), (x.double() ** 2).sum()]),
                     "x_head": x.flatten()[:16].clone(), "logits": y, "labels": y.argmax(-1)}
        print(tag, y)
    torch.save(gold, os.path.join(OUT, "dcnn_level14_eval.pt"))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "level14":
        level14()
    else:
        main()
