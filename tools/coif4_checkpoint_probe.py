"""Second plausibility check on reference-held data, for coif4 (task 8) -- NEGATIVE RESULT, kept as a tool.

Run in the build container only (reads the reference's shipped checkpoint as data; nothing of it is copied):
    python3 tools/coif4_checkpoint_probe.py > profiles/r05_coif4_checkpoint_probe.txt

`/root/reference/models/model_packetscoif4_*.pt` was trained on ptwt's coif4 level-8 log-packets, but by an older layer
list than the reference's current DCNN: `cnn.0 .. cnn.16` without gaps (conv, PReLU, BatchNorm x 5, conv, PReLU), so
the three 2x2 max-pools are not modules of that Sequential and their placement is unknown.  Every one of the C(6,3) = 20
placements gives the flat size 320 the checkpoint's `fc.1.weight [2, 320]` needs (the first convolution pads by 2, the
others keep the size), so the geometry does not single one out.  This script builds all 20 (eval-mode BatchNorm with the
checkpoint's running statistics; pool after or before the PReLU -- two slopes of the checkpoint are negative, so the two
differ) and feeds them the oracle's coif4 level-8 log-packets of the reference's own recordings
(tests/golden/ref_wavs_lj008_0217.pt: 7 real / 42 re-synthesised one-second frames), normalised with the fixture's own
mean / std as tests/test_reference_audio.py does for sym5.

Outcome: NO placement separates the recordings at the decision threshold -- all 20 call every frame real, with logit
margins of -40 ... -80: the fixture's own statistics are not the training set's, and unlike the sym5 checkpoint this
one does not forgive that.  Rank statistics are inconclusive as well: the margin (fake - real logit) orders the frames
with an AUC of 0.78-0.92 in frequency (Gray) order against 0.70-0.88 in natural order, no placement standing out, and
rescaling the input by 0.5 / 0.25 moves the operating point from "all real" to "all fake" without a separating band
in between.  So the coif4 checkpoint gives no evidence for (or against) the coif4 taps / pad geometry; the WPT oracle
stays "parity unpinned" at the ptwt boundary and the sym5 checkpoint remains the only reference-held plausibility
check.  No test was added.
"""
import itertools, os, sys, time
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'audiodeepfake-detection_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, _p)
from oracle import torch_ref, wpt_oracle
from test_reference_audio import _frames, _natural_order
torch.set_num_threads(8)
ck = torch.load('/root/reference/models/model_packetscoif4_none_220_22050_22050_256_1-11025_0.7_0.0004_0.001_128_2_10e_DCNN_signsFalse_augcFalse_augnFalse_power2.0_fbmelgan_1secs_0.pt', map_location='cpu')
sd = {k.replace('module.', ''): v for k, v in ck['MODEL_STATE'].items()}
x, fake = _frames()
feats, _ = torch_ref.packets_torch(x, wpt_oracle.TAPS['coif4'], 8, log_scale=True)
print(feats.shape)
convs = [0, 3, 6, 9, 12, 15]
def bn(v, pre):
    return (v - sd[pre + '.running_mean'][None, :, None, None]) / torch.sqrt(sd[pre + '.running_var'][None, :, None, None] + 1e-5)
def forward(f, pools):
    v = f.permute(0, 1, 3, 2)
    for i, c in enumerate(convs):
        w = sd[f'cnn.{c}.weight']
        pad = 2 if c == 0 else (0 if w.shape[-1] == 1 else 1)
        v = F.conv2d(v, w, sd[f'cnn.{c}.bias'], padding=pad)
        v = F.prelu(v, sd[f'cnn.{c+1}.weight'])
        if i in pools:
            v = F.max_pool2d(v, 2)
        if c != 15:
            v = bn(v, f'cnn.{c+2}')
    v = v.permute(0, 2, 1, 3).contiguous()
    for j, (k, p, d) in zip((0, 3, 6), ((3, 1, 1), (5, 2, 2), (7, 2, 4))):
        pre = f'dil_conv.{j}'
        v = bn(v, pre) * sd[pre + '.weight'][None, :, None, None] + sd[pre + '.bias'][None, :, None, None]
        v = F.conv2d(v, sd[f'dil_conv.{j+1}.weight'], sd[f'dil_conv.{j+1}.bias'], padding=p, dilation=d)
        v = F.prelu(v, sd[f'dil_conv.{j+2}.weight'])
    v = v.flatten(2)
    v = F.linear(v, sd['fc.1.weight'], sd['fc.1.bias'])
    return v.mean(1)
mean, std = feats.mean(), feats.std()
fn = ((feats - mean) / std).contiguous()
fnat = ((_natural_order(feats) - mean) / std).contiguous()
print('slopes', [float(sd[f'cnn.{c+1}.weight']) for c in convs])
res = []
for pools in itertools.combinations(range(6), 3):
    t0 = time.time()
    with torch.no_grad():
        try:
            p = forward(fn, pools).argmax(-1) != 0
            q = forward(fnat, pools).argmax(-1) != 0
        except Exception as e:
            print(pools, 'ERR', e); continue
    ok = (p == fake)
    print(pools, 'correct', int(ok.sum()), 'real ok', int(ok[~fake].sum()), '/7 fake ok', int(ok[fake].sum()), '/42 | natural order: fake called fake', int(q[fake].sum()), 'real called real', int((~q[~fake]).sum()), f'{time.time()-t0:.1f}s', flush=True)

print("---- variants")
def forward2(f, pools, pool_first=False, first_pad=2):
    v = f.permute(0, 1, 3, 2)
    for i, c in enumerate(convs):
        w = sd[f'cnn.{c}.weight']
        pad = first_pad if c == 0 else (0 if w.shape[-1] == 1 else 1)
        v = F.conv2d(v, w, sd[f'cnn.{c}.bias'], padding=pad)
        if pool_first and i in pools:
            v = F.max_pool2d(v, 2)
        v = F.prelu(v, sd[f'cnn.{c+1}.weight'])
        if (not pool_first) and i in pools:
            v = F.max_pool2d(v, 2)
        if c != 15:
            v = bn(v, f'cnn.{c+2}')
    v = v.permute(0, 2, 1, 3).contiguous()
    if v.shape[1] != 13:
        return None
    for j, (k, p, d) in zip((0, 3, 6), ((3, 1, 1), (5, 2, 2), (7, 2, 4))):
        pre = f'dil_conv.{j}'
        v = bn(v, pre) * sd[pre + '.weight'][None, :, None, None] + sd[pre + '.bias'][None, :, None, None]
        v = F.conv2d(v, sd[f'dil_conv.{j+1}.weight'], sd[f'dil_conv.{j+1}.bias'], padding=p, dilation=d)
        v = F.prelu(v, sd[f'dil_conv.{j+2}.weight'])
    v = v.flatten(2)
    if v.shape[-1] != 320:
        return None
    return F.linear(v, sd['fc.1.weight'], sd['fc.1.bias']).mean(1)
best = []
with torch.no_grad():
    for pools in [(0, 2, 5), (0, 1, 2), (0, 2, 3), (2, 4, 5)]:
        for pf in (False, True):
            for ms in (0.0, -2.0, 2.0):
                for sc in (1.0, 0.5, 2.0):
                    o = forward2((fn + ms) * sc, pools, pf)
                    if o is None: continue
                    p = o.argmax(-1) != 0
                    ok = int((p == fake).sum())
                    best.append((ok, pools, pf, ms, sc, float((o[:,1]-o[:,0])[fake].mean()), float((o[:,1]-o[:,0])[~fake].mean())))
best.sort(reverse=True)
for b in best[:12]: print(b)
# margin statistics for the reference-like placement
o = forward2(fn, (0, 2, 5))
d = (o[:, 1] - o[:, 0])
print('margin fake mean %.3f min %.3f max %.3f | real mean %.3f' % (d[fake].mean(), d[fake].min(), d[fake].max(), d[~fake].mean()))

print("---- AUC")
def auc(d):
    f, r = d[fake], d[~fake]
    return float((f[:, None] > r[None, :]).float().mean())
with torch.no_grad():
    for pools in itertools.combinations(range(6), 3):
        row = []
        for sc in (1.0, 0.5, 0.25):
            o = forward2(fn * sc, pools); on = forward2(fnat * sc, pools)
            d = o[:, 1] - o[:, 0]; dn = on[:, 1] - on[:, 0]
            # best threshold accuracy
            thr = sorted(d.tolist()); acc = max(int(((d > t) == fake).sum()) for t in thr + [thr[0] - 1])
            row.append(f"sc {sc}: auc {auc(d):.3f} nat {auc(dn):.3f} bestacc {acc} acc0 {int(((d>0)==fake).sum())}")
        print(pools, ' | '.join(row), flush=True)
