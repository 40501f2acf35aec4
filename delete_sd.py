"""``_target_: delete_sd.DeleteSD`` resolves here (config/delete_sd.yaml task._target_)."""
from siss_amd.tasks import DeleteSD  # noqa: F401
