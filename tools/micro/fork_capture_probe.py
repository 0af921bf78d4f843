"""Does multi-stream (fork / join) HIP-graph capture work on this stack?  1: plain ops on the capturing thread; 2: the fork
inside an autograd backward (runs on the engine's device thread)."""
import sys, torch
dev = torch.device("cuda")
a = torch.randn(1024, 1024, device=dev); b = torch.randn(1024, 1024, device=dev)
side = torch.cuda.Stream()
cap = torch.cuda.Stream()

def fork_join(x, y):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        u = x @ y
    v = y @ x
    main.wait_stream(side)
    return u + v

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if stage == 1:
    fork_join(a, b); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        out = fork_join(a, b)
    g.replay(); torch.cuda.synchronize()
    print("stage 1 ok", float((out - (a @ b + b @ a)).abs().max()))
else:
    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, y):
            ctx.save_for_backward(x, y)
            return x @ y
        @staticmethod
        def backward(ctx, g):
            x, y = ctx.saved_tensors
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                gy = x.t() @ g
            gx = g @ y.t()
            main.wait_stream(side)
            return gx, gy
    x = a.clone().requires_grad_(True); y = b.clone().requires_grad_(True)
    x.grad = torch.zeros_like(x); y.grad = torch.zeros_like(y)
    F.apply(x, y).sum().backward(); torch.cuda.synchronize()
    x.grad.zero_(); y.grad.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        F.apply(x, y).sum().backward()
    g.replay(); torch.cuda.synchronize()
    print("stage 2 ok", float(x.grad.abs().sum()), float(y.grad.abs().sum()))
