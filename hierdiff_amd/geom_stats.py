"""Node-count histogram of the GEOM-Drugs coarse-grained (fragment) dataset.

Data only: the (number of fragments -> molecule count) table the reference loads from
endiffusion/conf/analyze/GEOM.yaml (diffusion_qm9.py:114-115) to draw N for each sample.  Key order
is the YAML's insertion order; it defines the categorical index -> N mapping of DistributionNodes
(models/distributions.py:63-82) and therefore must be preserved.
"""

GEOM_FRAGMENT_HISTOGRAM = {
    26: 1649, 14: 25253, 17: 20706, 22: 6284, 11: 19351, 23: 4629, 12: 21924, 13: 24071,
    15: 24877, 9: 12105, 20: 11181, 18: 17219, 8: 8454, 10: 15819, 6: 2590, 16: 23530,
    19: 14266, 24: 3330, 21: 8593, 4: 428, 7: 5006, 25: 2433, 31: 213, 27: 1047,
    5: 1254, 29: 487, 33: 140, 32: 155, 37: 59, 28: 676, 57: 28, 40: 16,
    36: 63, 3: 120, 30: 295, 35: 97, 58: 15, 38: 50, 34: 89, 56: 21,
    42: 18, 48: 10, 49: 9, 54: 23, 51: 7, 70: 2, 55: 20, 47: 12,
    2: 6, 39: 29, 61: 2, 66: 2, 59: 8, 44: 20, 50: 6, 45: 13,
    53: 5, 43: 19, 79: 3, 41: 25, 46: 12, 83: 1, 52: 8, 1: 1,
    65: 2, 60: 2, 76: 1,
}
