"""which scaled_dot_product_attention backend is fastest for the tri-plane transformer's shapes (fp32, 16 heads x 48, 3072 tokens; cross: 77 keys)"""
import torch, torch.nn.functional as F, time
from torch.nn.attention import sdpa_kernel, SDPBackend
def bench(name, backends, Lk):
    q = torch.randn(1, 16, 3072, 48, device="cuda", requires_grad=True); k = torch.randn(1, 16, Lk, 48, device="cuda", requires_grad=True); v = torch.randn(1, 16, Lk, 48, device="cuda", requires_grad=True)
    try:
        with sdpa_kernel(backends):
            for _ in range(3):
                o = F.scaled_dot_product_attention(q, k, v); o.sum().backward()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                o = F.scaled_dot_product_attention(q, k, v); o.sum().backward()
            torch.cuda.synchronize(); print(f"Lk={Lk} {name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms fwd+bwd")
    except Exception as e:
        print(f"Lk={Lk} {name}: failed ({str(e)[:80]})")
for Lk in (3072, 77):
    bench("flash", [SDPBackend.FLASH_ATTENTION], Lk)
    bench("efficient", [SDPBackend.EFFICIENT_ATTENTION], Lk)
    bench("math", [SDPBackend.MATH], Lk)
