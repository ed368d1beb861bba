"""Camera matrices in the convention this build owns.

sgl::Camera is not vendored in the reference, so the build defines: right-handed view space looking
down -z (glm::lookAtRH layout), Vulkan-style projection with depth in [0, 1] and a y-flip
(proj[1][1] < 0), so that image row 0 / ndc.y = -1 is the TOP of the picture.  Matrices are float32,
column-major (GLM layout): flat index = col * 4 + row.  The shaders only consume viewMatrix,
projectionMatrix and their inverses (Data/Shaders/Renderers/LineUniformData.glsl:28-31).
"""
import numpy as np

# Default test camera of the reference's headless harness: test/VolumetricPathTracingTestRenderer.cpp:34-41
DEFAULT_POSITION = (0.0, 0.0, 0.8)
DEFAULT_FOVY = float(np.float32(2.0 * np.arctan(np.float32(0.5))))
DEFAULT_NEAR = 0.01
DEFAULT_FAR = 100.0


def look_at(eye, center=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)):
    eye = np.asarray(eye, dtype=np.float64)
    center = np.asarray(center, dtype=np.float64)
    up = np.asarray(up, dtype=np.float64)
    f = center - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4, dtype=np.float64)  # m[row, col]
    m[0, :3] = s
    m[1, :3] = u
    m[2, :3] = -f
    m[0, 3] = -np.dot(s, eye)
    m[1, 3] = -np.dot(u, eye)
    m[2, 3] = np.dot(f, eye)
    return np.ascontiguousarray(m.T.reshape(16), dtype=np.float32)  # column-major


def perspective(fovy, aspect, near=DEFAULT_NEAR, far=DEFAULT_FAR):
    t = np.tan(np.float64(fovy) / 2.0)
    m = np.zeros((4, 4), dtype=np.float64)  # m[row, col]
    m[0, 0] = 1.0 / (aspect * t)
    m[1, 1] = -1.0 / t  # y flip: row 0 of the image is the top
    m[2, 2] = far / (near - far)
    m[3, 2] = -1.0
    m[2, 3] = -(far * near) / (far - near)
    return np.ascontiguousarray(m.T.reshape(16), dtype=np.float32)


def default_camera(width, height, position=DEFAULT_POSITION, fovy=DEFAULT_FOVY):
    """(view, proj, fovy, near, far) of the reference's fixed test camera for a width x height viewport."""
    return (look_at(position), perspective(fovy, float(width) / float(height)), fovy, DEFAULT_NEAR, DEFAULT_FAR)
