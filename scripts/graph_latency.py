"""Experiment: replay the one-image vision tower from a HIP graph captured through torch.cuda.graph
(the C-ABI enqueues on torch's current stream and makes no forbidden call once its workspaces exist)."""
import sys, time
sys.path.insert(0, ".")
import torch
import tvc_amd as pkg
arch = pkg.get_arch("ViT-L/14")
w = pkg.synth.make_clip_weights(arch, seed=0)
eng = pkg.TVCEngine(arch, w[0], w[1], device="cuda:0")
for B in (1, 8):
    img = pkg.synth.make_images(B, arch.image_size, seed=1).cuda()
    for _ in range(3): ref = eng.encode_image(img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out = eng.encode_image(img)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 20 * 1e3
    static_in = img.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): eng.encode_image(static_in)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        static_out = eng.encode_image(static_in)
    static_in.copy_(img); g.replay(); torch.cuda.synchronize()
    ok = torch.equal(static_out, ref)
    t0 = time.perf_counter()
    for _ in range(20): static_in.copy_(img); g.replay()
    torch.cuda.synchronize(); graphed = (time.perf_counter() - t0) / 20 * 1e3
    print(f"B={B}: encode_image eager {eager:.2f} ms, graph replay {graphed:.2f} ms, identical={ok}")
