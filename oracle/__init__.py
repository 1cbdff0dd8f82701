"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain torch fp32 / numpy) of the
reference's hot path.  Nothing under ``stlpose_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and there only as the checker.

Parity status: PINNED for HRNet / PersonMSELoss / forward_pass / flip_back /
get_max_preds_hrnet / get_final_preds_hrnet / oks_nms against outputs of the
reference itself (``tests/golden/make_golden.py`` imports ``/root/reference/src``
in the build container and writes ``tests/golden/*.npz``).  UNPINNED for
VGGPerceptualLoss (reference needs torchvision + a weights download, neither
available), ``accuracy`` (reference line metrics.py:355-356 is corrupted) and
COCOeval (un-vendored pycocotools==2.0.0).
"""
