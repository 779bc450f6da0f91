"""CPU restatement of the reference's discriminator path (oracle; test-only): a functional CoModDiscriminator
driven by a state dict, built from plain aten ops and the restated upfirdn2d / bias_act of oracle/aten_ops.py.

Follows (reference tree) CMG = models/networks/CoModGAN:
  conv2d_resample        CMG/torch_utils/ops/conv2d_resample.py:57-155 (the branches the discriminator takes: 1x1 + down,
                         3x3 + down -> blur then strided conv, plain conv)
  Conv2dLayer            CMG/layers.py:115-162         FullyConnectedLayer   CMG/layers.py:81-111
  DiscriminatorBlock     CMG/generator.py:613-692 ('resnet' architecture)   MinibatchStdLayer   :696-718
  DiscriminatorEpilogue  CMG/generator.py:722-776      CoModDiscriminator    :780-836
  MappingNetwork (z_dim = 0, the label path of a conditional D)   CMG/layers.py:540-609
Pinned by tests/golden/D*.npz (tools/gen_golden_disc.py: logits, loss gradients, the R1 double backward).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import aten_ops as ops


def setup_filter(f):
    """upfirdn2d.setup_filter([1,3,3,1]): normalised, kept separable (1-D) (CMG/torch_utils/ops/upfirdn2d.py:65-111)."""
    f = torch.as_tensor(f, dtype=torch.float32)
    return f / f.sum()


def conv2d_resample(x, w, f=None, down=1, padding=0, up=1, flip_weight=True):
    """conv2d_resample.py:57-155.  Down only: the branches the discriminator takes (1x1: decimate then convolve; 3x3: blur then strided
    conv).  With up > 1: the DEFINITION the reference states as its generic path (:151-155) -- zero-insert upsampling through the
    low-pass filter (gain up^2), the convolution (true convolution when flip_weight is False, as Conv2dLayer asks for up > 1,
    layers.py:156), decimation -- not the transposed-convolution decomposition the reference (and the product) actually execute:
    the golden C1 pins that the two agree."""
    kh, kw = int(w.shape[2]), int(w.shape[3])
    fw = int(f.shape[-1]) if f is not None else 1
    px0 = px1 = py0 = py1 = int(padding)
    if up > 1:
        px0 += (fw + up - 1) // 2
        px1 += (fw - up) // 2
        py0 += (fw + up - 1) // 2
        py1 += (fw - up) // 2
        if down > 1:
            px0 += (fw - down + 1) // 2
            px1 += (fw - down) // 2
            py0 += (fw - down + 1) // 2
            py1 += (fw - down) // 2
        x = ops.upfirdn2d(x, f, up=up, padding=[px0, px1, py0, py1], gain=up ** 2)
        x = F.conv2d(x, w if flip_weight else w.flip([2, 3]))
        return ops.upfirdn2d(x, f, down=down) if down > 1 else x
    if down > 1:
        px0 += (fw - down + 1) // 2
        px1 += (fw - down) // 2
        py0 += (fw - down + 1) // 2
        py1 += (fw - down) // 2
    if kw == 1 and kh == 1 and down > 1:                      # :118-122  downsample first, then 1x1
        x = ops.upfirdn2d(x, f, down=down, padding=[px0, px1, py0, py1])
        return F.conv2d(x, w)
    if down > 1:                                              # :130-134  blur, then strided conv
        x = ops.upfirdn2d(x, f, padding=[px0, px1, py0, py1])
        return F.conv2d(x, w, stride=down)
    assert px0 == px1 == py0 == py1 and px0 >= 0              # :151-153
    return F.conv2d(x, w, padding=px0)


def conv2d_layer(sd, prefix, x, kernel_size, act='linear', down=1, gain=1.0, conv_clamp=None, filt=None, up=1):
    w = sd[prefix + 'weight']
    w = w * (1.0 / np.sqrt(w.shape[1] * kernel_size ** 2))                    # layers.py:137,154
    b = sd.get(prefix + 'bias')
    x = conv2d_resample(x, w, f=filt, down=down, padding=kernel_size // 2, up=up, flip_weight=(up == 1))
    act_gain = ops.ACTIVATIONS[act][2] * gain
    act_clamp = conv_clamp * gain if conv_clamp is not None else None
    return ops.bias_act(x, b, act=act, gain=act_gain, clamp=act_clamp)


def fully_connected(sd, prefix, x, act='linear', lr_multiplier=1.0):
    w = sd[prefix + 'weight']
    w = w * (lr_multiplier / np.sqrt(w.shape[1]))                              # layers.py:92,98
    b = sd.get(prefix + 'bias')
    if b is not None and lr_multiplier != 1:
        b = b * lr_multiplier                                                  # layers.py:93,101-102
    if act == 'linear' and b is not None:
        return torch.addmm(b.unsqueeze(0), x, w.t())
    return ops.bias_act(x.matmul(w.t()), b, act=act)


def minibatch_std(x, group_size, num_channels=1):
    n, c, h, w = x.shape
    g = min(group_size, n) if group_size is not None else n
    f = num_channels
    y = x.reshape(g, -1, f, c // f, h, w)
    y = y - y.mean(dim=0)
    y = y.square().mean(dim=0)
    y = (y + 1e-8).sqrt()
    y = y.mean(dim=[2, 3, 4])
    y = y.reshape(-1, f, 1, 1).repeat(g, 1, h, w)
    return torch.cat([x, y], dim=1)


def label_mapping(sd, c, num_layers=8, lr_multiplier=0.01):
    """MappingNetwork(z_dim=0, num_ws=None, w_avg_beta=None).forward(None, c) (layers.py:578-596): embed, normalise the second moment,
    ``num_layers`` lrelu FC layers."""
    y = fully_connected(sd, 'mapping.embed.', c.to(torch.float32))
    x = y * (y.square().mean(dim=1, keepdim=True) + 1e-8).rsqrt()
    for idx in range(num_layers):
        x = fully_connected(sd, f'mapping.fc{idx}.', x, act='lrelu', lr_multiplier=lr_multiplier)
    return x


def discriminator(sd, img, img_resolution, mbstd_group_size=4, conv_clamp=None, c=None):
    """CoModDiscriminator.forward(img, c), architecture 'resnet', fp32; ``c`` given (and a ``mapping.*`` branch in the state dict):
    the conditional form -- the epilogue's cmap_dim outputs projected onto the mapped label (generator.py:771-773)."""
    filt = setup_filter([1, 3, 3, 1])
    log2 = int(np.log2(img_resolution))
    x = None
    for res in [2 ** i for i in range(log2, 2, -1)]:
        p = f'b{res}.'
        if x is None:
            x = conv2d_layer(sd, p + 'fromrgb.', img, 1, act='lrelu', conv_clamp=conv_clamp)
        y = conv2d_layer(sd, p + 'skip.', x, 1, down=2, gain=np.sqrt(0.5), filt=filt)
        x = conv2d_layer(sd, p + 'conv0.', x, 3, act='lrelu', conv_clamp=conv_clamp)
        x = conv2d_layer(sd, p + 'conv1.', x, 3, act='lrelu', down=2, gain=np.sqrt(0.5), conv_clamp=conv_clamp, filt=filt)
        x = y + x
    x = minibatch_std(x, mbstd_group_size)
    x = conv2d_layer(sd, 'b4.conv.', x, 3, act='lrelu', conv_clamp=conv_clamp)
    x = fully_connected(sd, 'b4.fc.', x.flatten(1), act='lrelu')
    x = fully_connected(sd, 'b4.out.', x)
    if 'mapping.embed.weight' in sd:
        cmap = label_mapping(sd, c)
        x = (x * cmap).sum(dim=1, keepdim=True) * (1 / np.sqrt(cmap.shape[1]))
    return x


def d_losses(sd, fake, real, img_resolution, lambda_r1=10.0, **kw):
    """The discriminator half of the step (models/comodgan_model.py:128-149): returns
    (loss_fake, loss_real, loss_r1, gen_logits, real_logits, r1_grads) with graphs attached."""
    gen_logits = discriminator(sd, fake, img_resolution, **kw)
    loss_fake = F.softplus(gen_logits).mean()
    real_tmp = real.detach().requires_grad_(True)
    real_logits = discriminator(sd, real_tmp, img_resolution, **kw)
    loss_real = F.softplus(-real_logits).mean()
    r1_grads, = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real_tmp], create_graph=True, only_inputs=True)
    loss_r1 = r1_grads.square().sum([1, 2, 3]).mean() * 0.5
    return loss_fake, loss_real, loss_r1, gen_logits, real_logits, r1_grads
