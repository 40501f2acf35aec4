"""``_target_: delete_tshirt.DeleteTShirt`` resolves here (config/delete_tshirt.yaml task._target_)."""
from siss_amd.tasks import DeleteTShirt  # noqa: F401
