"""init_seeds / select_device / fuse_conv_and_bn -- mirror of the reference's utils/torch_utils.py."""
import torch


def init_seeds(seed=0):
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def select_device(device='', apex=False):
    cpu_request = device.lower() == 'cpu'
    cuda = False if cpu_request else torch.cuda.is_available()
    if cuda:
        ng = torch.cuda.device_count()
        for i in range(ng):
            x = torch.cuda.get_device_properties(i)
            print("%sdevice%g _HipDeviceProperties(name='%s', total_memory=%dMB)" %
                  ('Using HIP ' if i == 0 else ' ' * 10, i, x.name, x.total_memory / 1024 ** 2))
    else:
        print('Using CPU')
    return torch.device('cuda:0' if cuda else 'cpu')


def fuse_conv_and_bn(conv, bn):
    """W' = diag(gamma/sqrt(var+eps)) W,  b' = b + beta - gamma*mean/sqrt(var+eps)   (torch_utils.py:45-69)."""
    with torch.no_grad():
        fusedconv = torch.nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size,
                                    stride=conv.stride, padding=conv.padding, bias=True).to(conv.weight.device)
        w_conv = conv.weight.clone().view(conv.out_channels, -1)
        w_bn = torch.diag(bn.weight.div(torch.sqrt(bn.eps + bn.running_var)))
        fusedconv.weight.copy_(torch.mm(w_bn, w_conv).view(fusedconv.weight.size()))
        b_conv = conv.bias if conv.bias is not None else torch.zeros(conv.weight.size(0), device=conv.weight.device)
        b_bn = bn.bias - bn.weight.mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps))
        fusedconv.bias.copy_(torch.mm(w_bn, b_conv.reshape(-1, 1)).reshape(-1) + b_bn)
        return fusedconv
