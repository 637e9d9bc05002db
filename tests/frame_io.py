"""OpenCV-FileStorage-XML frame files as the publisher writes them (camTopic2PublisherPyD.cpp:306-383:
framemono_%04d.xml holding mono_%d (8-bit) and depth_%d (16-bit) per pyramid level).  Test-side writer; the engine's
reader is dvo_amd::SolveDVO::loadFromFile (include/dvo_amd.hpp)."""
import numpy as np


def _matrix(name, a):
    dt = {np.dtype(np.uint8): "u", np.dtype(np.uint16): "w", np.dtype(np.float32): "f"}[a.dtype]
    rows, cols = a.shape
    flat = a.ravel()
    lines = []
    per = 20 if dt != "f" else 8
    for i in range(0, flat.size, per):
        lines.append("    " + " ".join(("%d" % v) if dt != "f" else ("%.8e" % v) for v in flat[i:i + per]))
    return ('<%s type_id="opencv-matrix">\n  <rows>%d</rows>\n  <cols>%d</cols>\n  <dt>%s</dt>\n  <data>\n%s</data></%s>\n'
            % (name, rows, cols, dt, "\n".join(lines), name))


def write_frame_xml(path, pyramid):
    """pyramid: list over levels of (mono8 (rows, cols) uint8, depth16 (rows, cols) uint16), row-major"""
    with open(path, "w") as f:
        f.write('<?xml version="1.0"?>\n<opencv_storage>\n')
        for i, (g, d) in enumerate(pyramid):
            f.write(_matrix("mono_%d" % i, np.ascontiguousarray(g, dtype=np.uint8)))
            f.write(_matrix("depth_%d" % i, np.ascontiguousarray(d, dtype=np.uint16)))
        f.write("</opencv_storage>\n")
