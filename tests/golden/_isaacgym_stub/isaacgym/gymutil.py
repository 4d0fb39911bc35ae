"""Names only (base_task.py imports these at module import time)."""


def _stub(*a, **k):
    raise RuntimeError("isaacgym stub")


get_property_setter_map = get_property_getter_map = get_default_setter_args = _stub
apply_random_samples = check_buckets = generate_random_samples = _stub
parse_arguments = parse_sim_config = _stub
