"""Optional pretrained weights for the two OUT networks: a torchvision-style ResNet-18 state_dict (or a reference
LASR checkpoint, whose trunk keys are encoder.resnet_conv.resnet.layerN.*) loads into the encoder trunk, a torchvision /
LPIPS AlexNet state_dict into the perceptual network.  The state_dicts are synthesised with the right names and shapes
(torchvision is not needed: the key scheme is the contract)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lasr_amd.nnutils import mesh_net                               # noqa: E402


def torchvision_resnet18_keys(trunk):
    """This trunk's tensors under torchvision's names."""
    out = {}
    for k, v in trunk.state_dict().items():
        parts = k.split('.')
        if parts[0] == 'layers':
            k = 'layer%d.%s' % (int(parts[1]) + 1, '.'.join(parts[2:]))
        out[k] = torch.randn_like(v.float()).to(v.dtype) if v.dtype.is_floating_point else v.clone()
    return out


@pytest.mark.parametrize('prefix', ['', 'encoder.resnet_conv.resnet.'])
def test_resnet18_state_dict_loads_under_both_naming_schemes(tmp_path, prefix):
    trunk = mesh_net.ResNetConv()
    tv = torchvision_resnet18_keys(trunk)
    tv['fc.weight'] = torch.zeros(1000, 512)                         # ignored
    path = str(tmp_path / 'r18.pth')
    torch.save({prefix + k: v for k, v in tv.items()}, path)
    n = mesh_net.load_resnet18_weights(trunk, path)
    assert n == len(trunk.state_dict())
    assert torch.equal(trunk.layers[1][0].downsample[0].weight, tv['layer2.0.downsample.0.weight'])
    assert torch.equal(trunk.conv1.weight, tv['conv1.weight'])
    assert mesh_net.map_resnet_key('fc.weight') is None and mesh_net.map_resnet_key('layer4.1.bn2.bias') == 'layers.3.1.bn2.bias'


def test_file_without_trunk_tensors_is_an_error(tmp_path):
    path = str(tmp_path / 'junk.pth')
    torch.save({'foo': torch.zeros(3)}, path)
    with pytest.raises(ValueError):
        mesh_net.load_resnet18_weights(mesh_net.ResNetConv(), path)


@pytest.mark.parametrize('scheme', ['features.%s', 'net.slice%d.%s'])
def test_alexnet_weights_torchvision_and_lpips_names(tmp_path, scheme):
    net = mesh_net.PerceptualDistance()
    sd = {}
    for j, (idx, conv) in enumerate(zip(('0', '3', '6', '8', '10'), net.convs)):
        name = scheme % idx if scheme.count('%') == 1 else scheme % (j + 1, idx)
        sd[name + '.weight'] = torch.randn_like(conv.weight)
        sd[name + '.bias'] = torch.randn_like(conv.bias)
    path = str(tmp_path / 'alex.pth')
    torch.save(sd, path)
    assert net.load_weights(path) == 10
    name = scheme % '6' if scheme.count('%') == 1 else scheme % (3, '6')
    assert torch.equal(net.convs[2].weight, sd[name + '.weight'])
    assert not any(p.requires_grad for p in net.parameters())       # stays frozen
    torch.save({k: v for k, v in sd.items() if '10' not in k}, path)
    with pytest.raises(ValueError):
        mesh_net.PerceptualDistance().load_weights(path)
