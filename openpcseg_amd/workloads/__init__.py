"""Callers of the hot path used for measurement: synthetic LiDAR scans and the MinkUNet
segmentor definition (the reference's own model code cannot travel to the GPU box)."""
