import sys, time, torch
sys.path.insert(0, '.')
from afcm_amd.networks_discriminator import CoModDiscriminator
import torch.nn.functional as F
nfp16 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
D = CoModDiscriminator(c_dim=0, img_resolution=256, img_channels=5, channel_base=16384, channel_max=512, num_fp16_res=nfp16, conv_clamp=(256 if nfp16 else None),
                       block_kwargs=dict(fp16_dtype=torch.bfloat16), epilogue_kwargs=dict(mbstd_group_size=16)).cuda()
x = torch.randn(16, 5, 256, 256, device='cuda')
def it():
    for p in D.parameters(): p.grad = None
    F.softplus(D(x, None)).mean().backward()
    xr = x.detach().requires_grad_(True)
    lg = D(xr, None)
    r1, = torch.autograd.grad(lg.sum(), xr, create_graph=True)
    (F.softplus(-lg).mean() + 5.0 * r1.square().sum([1,2,3]).mean()).backward()
for _ in range(3): it()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): it()
torch.cuda.synchronize(); print('num_fp16_res', nfp16, (time.time() - t0) / 5 * 1e3, 'ms per D update (batch 16)')
