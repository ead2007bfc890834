from . import vgg, optical_flow  # noqa: F401
from .vgg import vgg16, VGG16_Weights  # noqa: F401
