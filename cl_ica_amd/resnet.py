"""ResNet-18 in plain ``torch.nn`` with torchvision's module names.

The 3DIdent driver builds its encoder as ``torchvision.models.resnet18(False, num_classes=10 * n_latents)``
(/root/reference/main_3dident.py:287-292, 365-370).  torchvision is not part of the MI355X image, so ``threedident.setup_f`` falls
back to this module when the import fails: the same architecture (7x7 stem, max-pool, four stages of two basic blocks, global
average pool, fc) and the same attribute names -- ``conv1, bn1, layer1..4[.i].{conv1,bn1,conv2,bn2,downsample.{0,1}}, fc`` -- so a
state dict written by torchvision's model loads here and vice versa.  The convolutions run on PyTorch-ROCm / MIOpen, as BASELINE
config 4 prescribes ("conv path via PyTorch-ROCm + HIP loss"); nothing here is device code of ours.
"""
from __future__ import annotations

import torch
from torch import nn

__all__ = ["resnet18", "ResNet18", "BasicBlock"]


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int = 1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))


class ResNet18(nn.Module):
    def __init__(self, num_classes: int = 1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(BasicBlock(64, 64, 1), BasicBlock(64, 64, 1))
        self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128, 1))
        self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256, 1))
        self.layer4 = nn.Sequential(BasicBlock(256, 512, 2), BasicBlock(512, 512, 1))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():                      # torchvision's initialisation (models/resnet.py)
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet18(pretrained: bool = False, num_classes: int = 1000, **_unused) -> ResNet18:
    """Call-compatible with ``torchvision.models.resnet18(pretrained, num_classes=...)`` as the reference uses it (weights are never
    downloaded here: ``pretrained=True`` raises)."""
    if pretrained:
        raise ValueError("no pretrained weights in this image (the reference trains from scratch: main_3dident.py:365 passes False)")
    return ResNet18(num_classes)
