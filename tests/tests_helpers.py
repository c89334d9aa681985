"""Shared helpers for the test-suite (skeletons, tolerances)."""
PARENTS = {
    17: [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15],            # reference reconstruction.py:95
    19: [-1, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 10, 13, 14, 10, 16, 17],  # reference reconstruction.py:87
    15: [-1, 0, 1, 2, 3, 1, 5, 6, 0, 8, 9, 0, 11, 12, 1],                    # reference common/humaneva_dataset.py:7
    16: [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 8, 10, 11, 8, 13, 14],               # reference h36m_dataset.py:267-277
}
