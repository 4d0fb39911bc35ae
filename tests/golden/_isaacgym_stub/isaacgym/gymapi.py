"""Names only: nothing here simulates anything."""


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()


def acquire_gym(*a, **k):
    raise RuntimeError("isaacgym stub: there is no simulator behind this name")


SimParams = Transform = Vec3 = Quat = AssetOptions = PlaneParams = CameraProperties = _Anything
UP_AXIS_Z = 1
UP_AXIS_Y = 0
DOF_MODE_NONE = 0
DOMAIN_SIM = 0
MESH_VISUAL = 0
LOCAL_SPACE = 0
KEY_ESCAPE = 0
KEY_V = 1
SIM_PHYSX = 1
SIM_FLEX = 0
