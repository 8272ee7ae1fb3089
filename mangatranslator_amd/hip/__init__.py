from .lib import MtxLibrary, get_library  # noqa: F401
from .plan import Act, Plan, PlanBuilder  # noqa: F401
