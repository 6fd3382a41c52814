"""Import alias: `import pointcloud_bridge_amd` loads the package kept in `pointcloud-bridge_amd/`.

The product directory carries the project's hyphenated name, which Python cannot import directly;
this stub points the package search path at it and runs its __init__.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "pointcloud-bridge_amd")
__path__ = [_REAL]
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
del _f
