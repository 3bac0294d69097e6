"""Would Winograd F(4x4,3x3) pass the error gate?  A CPU simulation in fp32 arithmetic (transforms, products summed over the
input channels and the output transform all in float32, weights transformed in float64 and rounded once) against the float64
direct convolution, on tools/conv_error.py's data (post-ReLU-like inputs x 1.5, He-scaled weights) for the layer3 / layer4
shapes - before any kernel is written.  The round-3 verdict's gate: max error <= 3x the shipped F(2x2,3x3) kernel's
(7.7e-6 on |y| ~ 7, profiles/r03_conv_error.txt) = 2.3e-5.

    python tools/wino_f4_error.py        (CPU only; ~1 min)

F(m x m, 3x3) by Toom-Cook with interpolation points p_0 .. p_{n-2} and infinity, n = m + 2:  y = A^T [(G g G^T) .* (B^T d B)] A with
A^T = (evaluation of a degree m-1 polynomial at the points)^T, G = evaluation of the filter polynomial, B^T = (V^-1)^T, V the
Vandermonde matrix of the points (infinity: the leading coefficient)."""
import torch
import torch.nn.functional as Fn

torch.manual_seed(0)


def matrices(points, m=4, r=3):
    n = m + r - 1
    V = torch.zeros(n, n, dtype=torch.float64)
    Eg, Eh = torch.zeros(n, r, dtype=torch.float64), torch.zeros(n, m, dtype=torch.float64)
    for j, p in enumerate(points):
        V[j] = torch.tensor([p ** k for k in range(n)], dtype=torch.float64)
        Eg[j] = torch.tensor([p ** k for k in range(r)], dtype=torch.float64)
        Eh[j] = torch.tensor([p ** k for k in range(m)], dtype=torch.float64)
    V[n - 1] = 0
    V[n - 1, n - 1] = 1
    Eg[n - 1] = 0
    Eg[n - 1, r - 1] = 1
    Eh[n - 1] = 0
    Eh[n - 1, m - 1] = 1
    return torch.linalg.inv(V).t().contiguous(), Eg, Eh.t().contiguous()        # B^T, G, A^T


def winograd(x, w, BT, G, AT, m):
    a = BT.shape[0]
    N, C, H, W = x.shape
    Co = w.shape[0]
    TY, TX = -(-H // m), -(-W // m)
    xp = Fn.pad(x, (1, TX * m - W + 1, 1, TY * m - H + 1))
    p = xp.unfold(2, a, m).unfold(3, a, m)
    V = torch.einsum("ij,nctujk,lk->nctuil", BT.float(), p, BT.float())
    U = torch.einsum("ij,ocjk,lk->ocil", G, w.double(), G).float()
    M = torch.einsum("ocil,nctuil->notuil", U, V)
    Y = torch.einsum("ij,notujk,lk->notuil", AT.float(), M, AT.float())
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, Co, TY * m, TX * m)[:, :, :H, :W]


F2 = (torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64),
      torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64),
      torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64))
POINTS = {"0 +-1 +-2 (the usual choice: additions and shifts only)": [0, 1, -1, 2, -2, 0],
          "0 +-1 +-1/2": [0, 1, -1, .5, -.5, 0],
          "0 +-1 2 -1/2": [0, 1, -1, 2, -.5, 0],
          "0 +-3/4 +-3/2": [0, .75, -.75, 1.5, -1.5, 0],
          "0 +-0.7 +-1.4 (best of those tried; needs multiplications)": [0, .7, -.7, 1.4, -1.4, 0]}
for name, C, H, W in (("layer3 256->256 50x84", 256, 50, 84), ("layer4 512->512 50x84", 512, 50, 84)):
    x = torch.randn(1, C, H, W).relu() * 1.5
    w = torch.randn(C, C, 3, 3) * (2.0 / (C * 9)) ** 0.5
    ref = Fn.conv2d(x.double(), w.double(), None, 1, 1)

    def err(y):
        e = (y.double() - ref).abs()
        return f"max {e.max().item():.2e} rms {e.pow(2).mean().sqrt().item():.2e}"

    print(f"{name}: |y|max {ref.abs().max().item():.2f}")
    print(f"   direct form in fp32                       {err(Fn.conv2d(x, w, None, 1, 1))}")
    print(f"   F(2x2,3x3)                                {err(winograd(x, w, *F2, 2))}")
    for label, pts in POINTS.items():
        print(f"   F(4x4,3x3) points {label:58s} {err(winograd(x, w, *matrices(pts), 4))}")
