"""Planning-world containers with the attribute names the reference planners read.

Mirrors ``path_planning_utils/rrt_env.py:1-20`` (2D) and
``path_planning_utils_3d/rrt_env_3d.py:1-11`` (3D) of the reference:
``x_range/y_range[/z_range]``, ``obs_circle/obs_rectangle`` (2D),
``obs_ball/obs_box`` (3D).  ``env_dims`` is (height, width[, depth]) i.e.
(y, x[, z]) extents, exactly like the reference unpacks it.
"""


class Env:
    """2D world: ``env_dict`` keys env_dims, circle_obstacles, rectangle_obstacles."""

    def __init__(self, env_dict):
        self.img_height, self.img_width = env_dict["env_dims"]
        self.x_range = (0, self.img_width)
        self.y_range = (0, self.img_height)
        h, w = self.img_height, self.img_width
        # one-pixel frame around the canvas, [x, y, w, h] each (only drawn, never collided with)
        self.obs_boundary = [[-1, -1, 1, h + 1], [-1, h, w + 1, 1], [0, -1, w + 1, 1], [w, 0, 1, h + 1]]
        self.obs_circle = env_dict["circle_obstacles"]
        self.obs_rectangle = env_dict["rectangle_obstacles"]


class Env3D:
    """3D world: ``env_dict`` keys env_dims, ball_obstacles, box_obstacles."""

    def __init__(self, env_dict):
        self.env_height, self.env_width, self.env_depth = env_dict["env_dims"]
        self.x_range = (0, self.env_width)
        self.y_range = (0, self.env_height)
        self.z_range = (0, self.env_depth)
        self.obs_ball = env_dict["ball_obstacles"]
        self.obs_box = env_dict["box_obstacles"]
