"""``jpeg.utils`` of the reference (src/jpeg/utils.py:24-41): the root-size helper, host-side integer arithmetic."""
from .tables import largest_power_of_2

__all__ = ["largest_power_of_2"]
