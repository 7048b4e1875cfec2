"""relativitypathtracer_amd — MI355X-native render path of the Relativity Path Tracer.

One data-parallel hot path (the reference's ``render_kernel``) as hand-written HIP for gfx950
behind a C-ABI (``include/rpt.h``), plus the host-side steps either side of it
(``include/rpt_scene.h``).  See DESIGN.md.
"""
from .scene import Scene, SceneError, write_png, write_ppm  # noqa: F401

__all__ = ["Scene", "SceneError", "write_png", "write_ppm"]
