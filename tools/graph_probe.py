"""Does a hipGraph of the LCNN bf16 evaluation forward pay at B = 128?  (development probe)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "audiodeepfake-detection_amd"))
import torch
import bench
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B in (128, 1024):
    args, trainer, _ = bench.build("stft-lcnn-eval-bf16", B, False, dev)
    batch = bench.synthetic_batch(B, 0, dev)
    trainer.model.eval()
    x = batch["audio"]
    def fwd():
        with torch.no_grad():
            return trainer.model(trainer._features(x))
    for _ in range(5): out = fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): out = fwd()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50 * 1e3
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fwd()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            gout = fwd()
        g.replay(); torch.cuda.synchronize()
        ok = torch.equal(gout, out)
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        graphed = (time.perf_counter() - t0) / 50 * 1e3
        print(f"B={B}: eager {eager:.3f} ms  graph replay {graphed:.3f} ms  same output: {ok}", flush=True)
    except Exception as e:
        print(f"B={B}: eager {eager:.3f} ms  capture failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
