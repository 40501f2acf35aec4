"""``_target_: delete_celeb.DeleteCeleb`` resolves here (config/delete_celeb.yaml task._target_)."""
from siss_amd.tasks import DeleteCeleb  # noqa: F401
