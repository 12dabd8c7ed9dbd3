"""
cuburn_amd — MI355X-native drop-in for cuburn's device path.

Mirrors the Python entry points of ``cuburn.render`` (``Renderer``, ``RenderManager``,
``Framebuffers.calc_dim``), ``cuburn.filters``, ``cuburn.profile`` and reads cuburn's
JSON genome / profile formats, over the C ABI of ``libflame_hip.so``
(include/flame_hip.h).  Only the hot path of SURVEY.md §8 lives here.
"""
__all__ = ['render', 'filters', 'profile', 'output', 'genome']
