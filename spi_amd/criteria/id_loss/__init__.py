from .id_loss import IDLoss  # noqa: F401
from .model_irse import Backbone  # noqa: F401
