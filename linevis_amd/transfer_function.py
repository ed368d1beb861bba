"""Transfer-function tables.

sgl::TransferFunctionWindow is not vendored in the reference; the build defines the texture as
N (default 256) RGBA float32 texels.  Colour points are given in sRGB, interpolated in linear RGB
(`interpolation_colorspace="Linear RGB"`, Data/TransferFunctions/Standard.xml) and stored back as
sRGB; opacity is interpolated linearly.  The renderer samples the table with linear filtering at
texel centres (i + 0.5) / N, clamp-to-edge (Data/Shaders/Utils/TransferFunction.glsl:66-71).
"""
import xml.etree.ElementTree as ET

import numpy as np

# Data/TransferFunctions/Standard.xml (values restated, not copied as a file)
STANDARD_COLOR_POINTS = [(0.0, (59, 76, 192)), (0.25, (144, 178, 254)), (0.5, (220, 220, 220)),
                         (0.75, (245, 156, 125)), (1.0, (180, 4, 38))]


def _srgb_to_linear(c):
    c = np.asarray(c, dtype=np.float64)
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def _linear_to_srgb(c):
    c = np.asarray(c, dtype=np.float64)
    return np.where(c <= 0.0031308, c * 12.92, 1.055 * np.power(np.maximum(c, 0.0), 1.0 / 2.4) - 0.055)


def build_table(color_points, opacity_points, n=256):
    xs = (np.arange(n, dtype=np.float64) + 0.0) / float(n - 1)
    cp = sorted(color_points)
    pos = np.array([p for p, _ in cp], dtype=np.float64)
    cols = _srgb_to_linear(np.array([c for _, c in cp], dtype=np.float64) / 255.0)
    rgb = np.stack([np.interp(xs, pos, cols[:, k]) for k in range(3)], axis=1)
    rgb = _linear_to_srgb(rgb)
    op = sorted(opacity_points)
    a = np.interp(xs, [p for p, _ in op], [o for _, o in op])
    return np.ascontiguousarray(np.concatenate([rgb, a[:, None]], axis=1), dtype=np.float32)


def standard(n=256, opacity=((0.0, 1.0), (1.0, 1.0))):
    """Standard.xml colours; opaque by default (configs 2/3/5), opacity ramp 0.1 -> 0.6 for config 4."""
    return build_table(STANDARD_COLOR_POINTS, list(opacity), n)


def standard_transparent(n=256):
    return standard(n, opacity=((0.0, 0.1), (1.0, 0.6)))


def load_xml(path, n=256):
    """Reads the reference's transfer-function XML format (OpacityPoints / ColorPoints)."""
    root = ET.parse(path).getroot()
    ops = [(float(e.get("position")), float(e.get("opacity"))) for e in root.iter("OpacityPoint")]
    cps = [(float(e.get("position")), (int(e.get("r")), int(e.get("g")), int(e.get("b"))))
           for e in root.iter("ColorPoint")]
    return build_table(cps, ops, n)
