"""Import-time stub (torchvision is absent offline); only names are needed."""
from . import transforms  # noqa: F401
