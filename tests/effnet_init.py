"""Seeded initialisation for EfficientNet parity tests: torch's default init drives an untrained 26-block MBConv stack to
1e-11 by the last endpoint, which would make an absolute 1e-4 tolerance vacuous.  Gains chosen so that every endpoint stays
O(0.3 - 1) (checked on B0 and B3); BatchNorm statistics and affine parameters are randomised so the folding is exercised."""
import torch, math
def init_effnet(net, seed, g_act=3.2, g_proj=0.6):
    gen = torch.Generator().manual_seed(seed)
    for name, m in net.named_modules():
        if isinstance(m, torch.nn.Conv2d):
            fan = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
            if "_se_" in name:
                std = (1.0 / fan) ** 0.5
            elif "_project" in name:
                std = (g_proj / fan) ** 0.5
            else:
                std = (g_act / fan) ** 0.5
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=gen) * std)
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.2)
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=gen) * 0.5 + 0.75)
                m.weight.copy_(torch.rand(m.num_features, generator=gen) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.num_features, generator=gen) * 0.1)
