"""CPU oracle for the autolabel NeRF hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``autolabel_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.

Pinning status
--------------
* ray-generation half (``raygen_oracle``): PINNED against golden vectors made
  by importing the reference's ``autolabel/dataset.py`` (tests/golden/).
* NeRF half (``nerf_oracle``): **parity unpinned** -- the reference's
  arithmetic lives in ``tinycudann`` (unpinned pip-from-git) and the
  ``ethz-asl/torch-ngp`` fork (empty submodule, no commit pin), neither of which
  is present under /root/reference, and the reference holds no test or golden
  vector for that half.  The oracle restates their published algorithms and is
  anchored on the reference's call sites (autolabel/models.py, trainer.py).
"""
