"""matplotlib renderers behind `planning(visualize=True)`, with the class names and `animation(...)` signatures of the
reference (path_planning_classes/rrt_visualizer_2d.py:11-28,102-118,141-191; path_planning_classes_3d/rrt_visualizer_3d.py:
8-34,139-155,182-231) so demo scripts that poke `planner.visualizer` keep working.  Plotting is outside the accelerated
path; the drawing code is this repo's own: tree edges as one LineCollection, the informed set as a
parametric ellipse / wire-frame ellipsoid built from the same (x_center, c_best, c_min, theta | C) the sampler uses.

    RRTStarVisualizer(x_start, x_goal, env).animation(vertices, vertex_parents, path, figure_title, animation=False,
                                                       img_filename=None, img_folder='visualization/planning_demo')
    IRRTStarVisualizer(...).animation(vertices, vertex_parents, path, figure_title, x_center, c_best, dist, theta,
                                      img_filename=None, img_folder=...)          (3D: rotation matrix C instead of theta)
    NRRTStarPNGVisualizer / NIRRTStarVisualizer: + set_path_point_cloud_pred / set_path_point_cloud_other
`img_filename=None` shows the figure (a no-op under the Agg backend), otherwise it is saved under img_folder.
"""
import math
import os

import numpy as np

DEFAULT_FOLDER = "visualization/planning_demo"


def _plt():
    import matplotlib
    if not os.environ.get("DISPLAY") and matplotlib.get_backend().lower() not in ("agg",):
        matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    return plt


class _Canvas:
    """shared figure handling: open, decorate, finish (show or save)"""
    dim = 2

    def __init__(self, x_start, x_goal, env):
        self.x_start = np.asarray(x_start, dtype=np.float64).ravel()
        self.x_goal = np.asarray(x_goal, dtype=np.float64).ravel()
        self.env = env
        self.fig = None
        self.ax = None

    def _open(self, figure_title):
        plt = _plt()
        self.fig = plt.figure(figsize=(6, 6))
        self.ax = self.fig.add_subplot(111, projection="3d" if self.dim == 3 else None)
        self.plot_grid(figure_title)
        return plt

    def _finish(self, plt, img_filename, img_folder):
        if img_filename is None:
            plt.show()
        else:
            os.makedirs(img_folder, exist_ok=True)
            self.fig.savefig(os.path.join(img_folder, img_filename), dpi=120)
        plt.close(self.fig)
        self.fig = self.ax = None

    def plot_start_goal(self):
        self.ax.plot(*[[c] for c in self.x_start], "s", color="tab:blue", ms=6)
        self.ax.plot(*[[c] for c in self.x_goal], "s", color="tab:red", ms=6)

    def plot_path(self, path):
        if path is not None and len(path):
            p = np.asarray(path)
            self.ax.plot(*[p[:, k] for k in range(self.dim)], "-", color="tab:red", lw=2)

    def plot_points(self, points, color, size=2):
        if points is not None and len(points):
            p = np.asarray(points)
            self.ax.scatter(*[p[:, k] for k in range(self.dim)], s=size, c=color)

    def plot_scene_path(self, path, figure_title, img_filename=None, img_folder=DEFAULT_FOLDER):
        plt = self._open(figure_title)
        self.plot_path(path)
        self._finish(plt, img_filename, img_folder)


class RRTStarVisualizer(_Canvas):
    dim = 2

    def plot_grid(self, figure_title):
        from matplotlib.patches import Circle, Rectangle
        ax = self.ax
        for (x, y, w, h) in getattr(self.env, "obs_boundary", []):
            ax.add_patch(Rectangle((x, y), w, h, fc="black", ec="black"))
        for (x, y, w, h) in self.env.obs_rectangle:
            ax.add_patch(Rectangle((x, y), w, h, fc="gray", ec="black"))
        for (x, y, r) in self.env.obs_circle:
            ax.add_patch(Circle((x, y), r, fc="gray", ec="black"))
        self.plot_start_goal()
        ax.set_xlim(self.env.x_range)
        ax.set_ylim(self.env.y_range)
        ax.set_aspect("equal")
        ax.set_title(figure_title)

    def plot_visited(self, vertices, vertex_parents, animation=False):
        from matplotlib.collections import LineCollection
        v = np.asarray(vertices)
        p = np.asarray(vertex_parents).astype(int)
        if len(v) > 1:
            seg = np.stack([v[1:], v[p[1:]]], axis=1)
            self.ax.add_collection(LineCollection(seg, colors="tab:green", linewidths=0.4))

    def animation(self, vertices, vertex_parents, path, figure_title, animation=False, img_filename=None,
                  img_folder=DEFAULT_FOLDER):
        plt = self._open(figure_title)
        self.plot_visited(vertices, vertex_parents, animation)
        self.plot_path(path)
        self._finish(plt, img_filename, img_folder)


class IRRTStarVisualizer(RRTStarVisualizer):
    def draw_ellipse(self, x_center, c_best, dist, theta):
        """informed set {x : |x - start| + |x - goal| <= c_best}: semi-axes c_best/2 along the start-goal direction
        (angle theta) and sqrt(c_best^2 - dist^2)/2 across, centred at x_center"""
        rad = c_best ** 2 - dist ** 2
        major, minor = c_best / 2.0, math.sqrt(rad if rad >= 0 else 1e-6) / 2.0
        t = np.linspace(0.0, 2.0 * math.pi, 181)
        cx, cy = float(np.ravel(x_center)[0]), float(np.ravel(x_center)[1])
        ex, ey = major * np.cos(t), minor * np.sin(t)
        ct, st = math.cos(theta), math.sin(theta)
        self.ax.plot(cx + ct * ex - st * ey, cy + st * ex + ct * ey, "--", color="darkorange", lw=1.5)

    def _overlay(self):
        pass

    def animation(self, vertices, vertex_parents, path, figure_title, x_center, c_best, dist, theta, img_filename=None,
                  img_folder=DEFAULT_FOLDER):
        plt = self._open(figure_title)
        self._draw_tree_layer(vertices, vertex_parents)
        self._overlay()
        if c_best != np.inf:
            self.draw_ellipse(x_center, c_best, dist, theta)
        self.plot_path(path)
        self._finish(plt, img_filename, img_folder)

    def _draw_tree_layer(self, vertices, vertex_parents):
        self.plot_visited(vertices, vertex_parents, False)


class NRRTStarPNGVisualizer(RRTStarVisualizer):
    def __init__(self, x_start, x_goal, env, path_point_cloud_pred=None):
        super().__init__(x_start, x_goal, env)
        self.path_point_cloud_pred = path_point_cloud_pred

    def set_path_point_cloud_pred(self, path_point_cloud_pred):
        self.path_point_cloud_pred = path_point_cloud_pred

    def animation(self, vertices, vertex_parents, path, figure_title, animation=False, img_filename=None,
                  img_folder=DEFAULT_FOLDER):
        plt = self._open(figure_title)
        self.plot_points(self.path_point_cloud_pred, "C1")
        self.plot_visited(vertices, vertex_parents, animation)
        self.plot_path(path)
        self._finish(plt, img_filename, img_folder)


class NIRRTStarVisualizer(IRRTStarVisualizer):
    """like the reference, the guided planners' picture shows the two point-cloud classes instead of the tree edges"""

    def __init__(self, x_start, x_goal, env, path_point_cloud_pred=None):
        super().__init__(x_start, x_goal, env)
        self.path_point_cloud_pred = path_point_cloud_pred
        self.path_point_cloud_other = None

    def set_path_point_cloud_other(self, path_point_cloud_other):
        self.path_point_cloud_other = path_point_cloud_other

    def set_path_point_cloud_pred(self, path_point_cloud_pred):
        self.path_point_cloud_pred = path_point_cloud_pred

    def _draw_tree_layer(self, vertices, vertex_parents):
        pass

    def _overlay(self):
        self.plot_points(self.path_point_cloud_pred, "C1")
        self.plot_points(self.path_point_cloud_other, "C0")


# ------------------------------------------------------------------------------------------------
# 3D
# ------------------------------------------------------------------------------------------------
class RRTStarVisualizer3D(_Canvas):
    dim = 3

    def plot_grid(self, figure_title):
        ax = self.ax
        u, w = np.meshgrid(np.linspace(0, 2 * math.pi, 16), np.linspace(0, math.pi, 10))
        for (x, y, z, r) in self.env.obs_ball:
            ax.plot_surface(x + r * np.cos(u) * np.sin(w), y + r * np.sin(u) * np.sin(w), z + r * np.cos(w),
                            color="gray", alpha=0.3, linewidth=0)
        for (x, y, z, bw, bh, bd) in self.env.obs_box:
            self._box(x, y, z, bw, bh, bd)
        self.plot_start_goal()
        ax.set_xlim(self.env.x_range)
        ax.set_ylim(self.env.y_range)
        ax.set_zlim(self.env.z_range)
        ax.set_title(figure_title)

    def _box(self, x, y, z, w, h, d):
        from mpl_toolkits.mplot3d.art3d import Poly3DCollection
        c = np.array([[x + a * w, y + b * h, z + e * d] for a in (0, 1) for b in (0, 1) for e in (0, 1)])
        quads = [[0, 1, 3, 2], [4, 5, 7, 6], [0, 1, 5, 4], [2, 3, 7, 6], [0, 2, 6, 4], [1, 3, 7, 5]]
        self.ax.add_collection3d(Poly3DCollection([c[q] for q in quads], facecolors="gray", edgecolors="k", linewidths=0.3,
                                                  alpha=0.15))

    def plot_visited(self, vertices, vertex_parents, animation=False):
        from mpl_toolkits.mplot3d.art3d import Line3DCollection
        v = np.asarray(vertices)
        p = np.asarray(vertex_parents).astype(int)
        if len(v) > 1:
            self.ax.add_collection3d(Line3DCollection(np.stack([v[1:], v[p[1:]]], axis=1), colors="tab:green", linewidths=0.3))

    def animation(self, vertices, vertex_parents, path, figure_title, animation=False, img_filename=None,
                  img_folder=DEFAULT_FOLDER):
        plt = self._open(figure_title)
        self.plot_visited(vertices, vertex_parents, animation)
        self.plot_path(path)
        self._finish(plt, img_filename, img_folder)


class IRRTStarVisualizer3D(RRTStarVisualizer3D):
    def draw_ellipsoid(self, x_center, c_best, dist, C):
        """wire-frame of C . diag(c_best/2, s, s) . unit sphere + x_center, s = sqrt(c_best^2 - dist^2)/2"""
        rad = c_best ** 2 - dist ** 2
        r = np.array([c_best / 2.0, math.sqrt(rad if rad >= 0 else 1e-6) / 2.0, math.sqrt(rad if rad >= 0 else 1e-6) / 2.0])
        u, w = np.meshgrid(np.linspace(0, 2 * math.pi, 24), np.linspace(0, math.pi, 13))
        ball = np.stack([np.cos(u) * np.sin(w), np.sin(u) * np.sin(w), np.cos(w)], axis=0).reshape(3, -1)
        pts = (np.asarray(C, dtype=np.float64).reshape(3, 3) @ (r[:, None] * ball)) + np.ravel(x_center)[:3, None]
        X, Y, Z = (pts[k].reshape(u.shape) for k in range(3))
        self.ax.plot_wireframe(X, Y, Z, color="darkorange", linewidth=0.4, alpha=0.6)

    def _draw_tree_layer(self, vertices, vertex_parents):
        self.plot_visited(vertices, vertex_parents, False)

    def _overlay(self):
        pass

    def animation(self, vertices, vertex_parents, path, figure_title, x_center, c_best, dist, C, img_filename=None,
                  img_folder=DEFAULT_FOLDER):
        plt = self._open(figure_title)
        self._draw_tree_layer(vertices, vertex_parents)
        self._overlay()
        if c_best != np.inf:
            self.draw_ellipsoid(x_center, c_best, dist, C)
        self.plot_path(path)
        self._finish(plt, img_filename, img_folder)


class NRRTStarPNGVisualizer3D(RRTStarVisualizer3D):
    def __init__(self, x_start, x_goal, env, path_point_cloud_pred=None):
        super().__init__(x_start, x_goal, env)
        self.path_point_cloud_pred = path_point_cloud_pred

    def set_path_point_cloud_pred(self, path_point_cloud_pred):
        self.path_point_cloud_pred = path_point_cloud_pred

    def animation(self, vertices, vertex_parents, path, figure_title, animation=False, img_filename=None,
                  img_folder=DEFAULT_FOLDER):
        plt = self._open(figure_title)
        self.plot_points(self.path_point_cloud_pred, "C1")
        self.plot_path(path)
        self._finish(plt, img_filename, img_folder)


class NIRRTStarVisualizer3D(IRRTStarVisualizer3D):
    def __init__(self, x_start, x_goal, env, path_point_cloud_pred=None):
        super().__init__(x_start, x_goal, env)
        self.path_point_cloud_pred = path_point_cloud_pred
        self.path_point_cloud_other = None

    def set_path_point_cloud_pred(self, path_point_cloud_pred):
        self.path_point_cloud_pred = path_point_cloud_pred

    def set_path_point_cloud_other(self, path_point_cloud_other):
        self.path_point_cloud_other = path_point_cloud_other

    def _draw_tree_layer(self, vertices, vertex_parents):
        pass

    def _overlay(self):
        self.plot_points(self.path_point_cloud_pred, "C1")


def file_stem(planner_name):
    """'NIRRT*-PNG(C) 2D' -> 'nirrt*-png(c)_2d': the reference's example file names ('irrt*_2d_example.png')"""
    return planner_name.lower().replace(" ", "_")
