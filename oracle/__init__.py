"""CPU oracle for the WVN feature_extractor -> traversability_estimator hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``wild_visual_navigation_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do,
and only as the checker / the reported CPU baseline -- never as the thing measured or shipped.

What it is: a plain PyTorch-CPU fp32 (and, for the integer/bit-exact pieces, numpy) restatement
of the reference algorithm for this path.  Every function cites the reference file:line it
follows.  Pin status (see DESIGN.md "Oracle"):

* first-party pieces (SimpleMLP, TraversabilityLoss, ConfidenceGenerator, Data/Batch,
  SegmentExtractor.{adjacency_list,centers}, FeatureExtractor.sparsify_features,
  MissionNode.update_supervision_signal, torch.optim.Adam trajectory) are PINNED: they were
  checked in the build container against the reference's own code imported from
  /root/reference (``oracle/pin_reference.py``) and the resulting input/output vectors are
  committed under ``tests/golden/``.
* the DINO ViT backbone and the STEGO head/cluster step live in the third-party package
  ``stego`` (leggedrobotics/self_supervised_segmentation, un-vendored, un-pinned, absent from
  /root/reference) and no reference test holds a golden vector for them: PARITY UNPINNED at
  that boundary.  The restatement follows the published DINO architecture and is guarded by a
  weight-copy cross-check against HuggingFace ``transformers.ViTModel`` (tests/test_oracle_vit.py).
"""
