from .aam_softmax import AngularAdditiveMarginSoftMaxLoss  # noqa: F401
from .binary_cross_entropy import BinaryCrossEntropyLoss  # noqa: F401
from .cross_entropy import CrossEntropyLoss  # noqa: F401
