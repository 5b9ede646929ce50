"""footprint_model parameter sets for getRobotFootprintFromParamServer (src/mpc_local_planner_ros.cpp:890-1001); make_ref_vectors.py records what the REFERENCE builds"""
SQUARE = [(0.2, 0.1), (-0.2, 0.1), (-0.2, -0.1), (0.2, -0.1)]


def cases():
    """name -> (footprint_model dictionary or None, costmap footprint or None, costmap pointer is null)"""
    c = {
        "none": (None, None, False), "point": ({"type": "point"}, None, False),
        "circular": ({"type": "circular", "radius": 0.3}, None, False), "circular_int": ({"type": "circular", "radius": 1}, None, False),
        "circular_missing": ({"type": "circular"}, None, False), "circular_text": ({"type": "circular", "radius": "0.3"}, None, False),
        "line": ({"type": "line", "line_start": [0.0, 0.0], "line_end": [0.4, 0.0]}, None, False), "line_ints": ({"type": "line", "line_start": [0, 0], "line_end": [1, 0]}, None, False),
        "line_missing_end": ({"type": "line", "line_start": [0.0, 0.0]}, None, False), "line_3d": ({"type": "line", "line_start": [0.0, 0.0, 0.0], "line_end": [0.4, 0.0]}, None, False),
        "two_circles": ({"type": "two_circles", "front_offset": 0.2, "front_radius": 0.25, "rear_offset": 0.1, "rear_radius": 0.2}, None, False),
        "two_circles_missing": ({"type": "two_circles", "front_offset": 0.2, "front_radius": 0.25}, None, False),
        "two_circles_ints": ({"type": "two_circles", "front_offset": 1, "front_radius": 1, "rear_offset": 0, "rear_radius": 1}, None, False),
        "polygon": ({"type": "polygon", "vertices": [[0.3, 0.2], [-0.3, 0.2], [-0.3, -0.2], [0.3, -0.2]]}, None, False),
        "polygon_ints": ({"type": "polygon", "vertices": [[1, 1], [-1, 1], [0, -1]]}, None, False),
        "polygon_two_points": ({"type": "polygon", "vertices": [[1.0, 1.0], [-1.0, 1.0]]}, None, False),
        "polygon_3d_point": ({"type": "polygon", "vertices": [[1.0, 1.0], [-1.0, 1.0], [0.0, -1.0, 0.0]]}, None, False),
        "polygon_text": ({"type": "polygon", "vertices": [[1.0, 1.0], [-1.0, 1.0], [0.0, "a"]]}, None, False),
        "polygon_flat_list": ({"type": "polygon", "vertices": [1.0, 2.0, 3.0]}, None, False), "polygon_missing": ({"type": "polygon"}, None, False),
        "unknown": ({"type": "blob"}, None, False),
        "costmap_2d": ({"type": "costmap_2d"}, SQUARE, False), "costmap_2d_without_costmap": ({"type": "costmap_2d"}, None, True),
    }
    return c
