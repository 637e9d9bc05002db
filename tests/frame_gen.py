"""Synthetic camera frames for the frame-path tests: the generator lives in the package (rgbd_odometry_amd/frame_gen.py)
because bench.py and the measurement tools use it too."""
from rgbd_odometry_amd.frame_gen import camera_frame  # noqa: F401
