"""CPU oracle for the SISS unlearning step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product
path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and there only as the checker / the timed
CPU baseline.  The product path (``siss_amd``) never imports this package and
fails loudly when the HIP library is missing.

Contents (each function cites the reference file:line it restates):

* ``schedule.py``   -- DDPM schedule + ``add_noise``           (diffusers 0.27.2, un-vendored)
* ``loss.py``       -- ``DDPMDeletionLoss`` restatement          (losses/ddpm_deletion_loss.py)
* ``unet.py``       -- torch-only ``UNet2DModel`` restatement    (diffusers 0.27.2, un-vendored)
* ``step.py``       -- one optimizer step of the delete_*.py loop (delete_celeb.py:557-773)

Pinning status
--------------
* loss + step: pinned against the reference's own ``losses/ddpm_deletion_loss.py``
  imported in the build container (``oracle/make_golden.py`` -> ``tests/golden/*.npz``).
* UNet: **parity unpinned** by the reference -- diffusers==0.27.2 is a pip
  dependency (environment.yml:232) that is neither vendored under /root/reference
  nor installed in this image.  The restatement follows the published
  architecture and is pinned only by the exact parameter count / tensor count /
  state-dict key names of google/ddpm-celebahq-256 (113,673,219 / 450).
"""
