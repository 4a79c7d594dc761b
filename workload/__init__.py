"""Synthetic workloads and reference-object stand-ins (TEST / BENCH INFRASTRUCTURE - not part of the product package).

* refmodel  - plain-Python stand-ins with the attributes of nhd.Node.Node / nhd.CfgTopology.CfgTopology that the hot path
              reads, built from the same NFD label dicts (used where the reference tree is absent: the GPU box)
* synth     - the seeded BASELINE clusters and pod mixes
* dist      - torch.distributed helpers of the multi-process bench / gloo tests
"""
