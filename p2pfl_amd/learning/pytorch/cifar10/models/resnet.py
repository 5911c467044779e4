"""CIFAR-10 ResNets (reference: fedstellar/learning/pytorch/cifar10/models/resnet.py:38-294).

``implementation="scratch"`` builds ResNet9 (conv3x3+BN+ReLU[+pool] blocks, resnet.py:179-198);
the reference keeps those blocks in a plain ``dict`` (resnet.py:198) so only the classifier is
registered, trained and federated — here they are proper sub-modules. The torchvision variants
(resnet18/34/50, resnet.py:200-205) are replaced by own BasicBlock/Bottleneck stacks of the same
shapes (torchvision is not available) with a ``fc -> out_channels`` head.
Optimizer: Adam(lr=1e-2, weight_decay=1e-4) (resnet.py:236-239).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .....models.torch_models import FederatedModule


def conv_block(cin, cout, pool=False):
    layers = [nn.Conv2d(cin, cout, kernel_size=3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]
    if pool:
        layers.append(nn.MaxPool2d(2))
    return nn.Sequential(*layers)


class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or cin != planes:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = torch.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return torch.relu(out + idt)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if stride != 1 or cin != planes * 4:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = torch.relu(self.bn1(self.conv1(x)))
        out = torch.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return torch.relu(out + idt)


class _ResNet(nn.Module):
    def __init__(self, block, layers, in_channels, num_classes):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(block, 64, layers[0], 1)
        self.layer2 = self._make(block, 128, layers[1], 2)
        self.layer3 = self._make(block, 256, layers[2], 2)
        self.layer4 = self._make(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)

    def _make(self, block, planes, n, stride):
        blocks = [block(self.inplanes, planes, stride)]
        self.inplanes = planes * block.expansion
        blocks += [block(self.inplanes, planes) for _ in range(n - 1)]
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = self.maxpool(torch.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


_VARIANTS = {"resnet18": (_BasicBlock, [2, 2, 2, 2]), "resnet34": (_BasicBlock, [3, 4, 6, 3]),
             "resnet50": (_Bottleneck, [3, 4, 6, 3])}


class CIFAR10ModelResNet(FederatedModule):
    def __init__(self, in_channels=3, out_channels=10, learning_rate=1e-2, metrics=None, confusion_matrix=None,
                 seed=None, implementation="scratch", classifier="resnet9"):
        super().__init__(in_channels, out_channels, learning_rate, metrics, confusion_matrix, seed)
        self.example_input_array = torch.rand(1, 3, 32, 32)
        self.implementation, self.classifier_name = implementation, classifier
        if implementation == "scratch" and classifier == "resnet9":
            self.conv1 = conv_block(in_channels, 64)
            self.conv2 = conv_block(64, 128, pool=True)
            self.res1 = nn.Sequential(conv_block(128, 128), conv_block(128, 128))
            self.conv3 = conv_block(128, 256, pool=True)
            self.conv4 = conv_block(256, 512, pool=True)
            self.res2 = nn.Sequential(conv_block(512, 512), conv_block(512, 512))
            self.classifier = nn.Sequential(nn.MaxPool2d(4), nn.Flatten(), nn.Linear(512, out_channels))
            self.backbone = None
        elif classifier in _VARIANTS:
            block, layers = _VARIANTS[classifier]
            self.backbone = _ResNet(block, layers, in_channels, out_channels)
        else:
            raise NotImplementedError(f"{implementation}/{classifier}")

    def forward(self, x):
        x = x.reshape(-1, 3, 32, 32)
        if self.backbone is not None:
            return self.backbone(x)
        out = self.conv2(self.conv1(x))
        out = self.res1(out) + out
        out = self.conv4(self.conv3(out))
        out = self.res2(out) + out
        return self.classifier(out)

    def configure_optimizers(self):
        return torch.optim.Adam(self.parameters(), lr=self.learning_rate, weight_decay=1e-4)
