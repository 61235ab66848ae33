def resnet18(*a, **k):
    raise RuntimeError("torchvision stub: the ResNet-18 audio branch is out of scope (SURVEY.md row 5)")
