from .resnet50 import ResNet50, ResNet50v2  # noqa: F401
