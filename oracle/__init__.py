"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's DeepLab-v3+ / active-selection path.

Nothing under oracle/ is part of the product: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it, and only as the checker.  The product (deep-active-semantic-segmentation_amd/)
never imports this package and fails loudly when libdass_hip.so is missing.

Parity status: PINNED.  Every function here was checked in the authoring container against the
reference's own modules imported from /root/reference (oracle/make_goldens.py, which also wrote the
fixtures under tests/golden/).  The reference itself never ships.
"""
