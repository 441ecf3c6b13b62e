import sys, time, torch
sys.path.insert(0, "/root/repo")
from smilify_amd import model_io, synthetic
t = model_io.load_model("/root/repo/data/models/SMILy_STICK.npz")
for B in (1, 8, 64):
    f = synthetic.make_problem(t, B, 1, 256, "cuda:0", window=10)
    f.begin_stage(5e-3)
    for _ in range(3): f.fit_step(synthetic.STAGE1_WEIGHTS, 100.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f.fit_step(synthetic.STAGE1_WEIGHTS, 100.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    # reference-style: forward + backward + torch Adam for one window
    opt = torch.optim.Adam([p for n, p in f.named_parameters() if p.requires_grad], lr=5e-3, betas=(0.5, 0.999))
    for _ in range(3):
        opt.zero_grad(); l, _ = f(list(range(B)), synthetic.STAGE1_WEIGHTS, 1); l.backward(); opt.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        opt.zero_grad(); l, _ = f(list(range(B)), synthetic.STAGE1_WEIGHTS, 1); l.backward(); opt.step()
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / 20
    print(f"B={B}: fit_step {dt*1e3:.2f} ms   forward+backward+torch.Adam {dt2*1e3:.2f} ms")
