"""Minimal matplotlib rendering for planning(visualize=True) (reference: rrt_visualizer_{2d,3d}.py,
plotting is outside the accelerated path - SURVEY.md §8f item 1)."""
import os


def draw_tree(planner, *args, figure_title=None, img_filename=None, **kwargs):
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception as e:  # pragma: no cover
        print("visualize: matplotlib unavailable (%s)" % e)
        return
    n = planner.num_vertices
    v, p = planner.vertices[:n], planner.vertex_parents[:n]
    fig = plt.figure(figsize=(6, 6))
    if planner.dim == 2:
        ax = fig.add_subplot(111)
        for (x, y, r) in planner.env.obs_circle:
            ax.add_patch(plt.Circle((x, y), r, color="gray"))
        for (x, y, w, h) in planner.env.obs_rectangle:
            ax.add_patch(plt.Rectangle((x, y), w, h, color="gray"))
        step = max(1, n // 4000)
        for i in range(1, n, step):
            ax.plot([v[i, 0], v[p[i], 0]], [v[i, 1], v[p[i], 1]], "-g", lw=0.3)
        pc = getattr(planner, "path_point_cloud_pred", None)
        if pc is not None and len(pc):
            ax.scatter(pc[:, 0], pc[:, 1], s=1, c="C1")
        if len(planner.path):
            ax.plot(planner.path[:, 0], planner.path[:, 1], "-r", lw=2)
        ax.plot(*planner.x_start, "bs")
        ax.plot(*planner.x_goal, "rs")
        ax.set_xlim(planner.x_range)
        ax.set_ylim(planner.y_range)
        ax.set_aspect("equal")
    else:
        ax = fig.add_subplot(111, projection="3d")
        step = max(1, n // 2000)
        for i in range(1, n, step):
            ax.plot([v[i, 0], v[p[i], 0]], [v[i, 1], v[p[i], 1]], [v[i, 2], v[p[i], 2]], "-g", lw=0.3)
        if len(planner.path):
            ax.plot(planner.path[:, 0], planner.path[:, 1], planner.path[:, 2], "-r", lw=2)
    ax.set_title(figure_title or "%s, iteration %d" % (planner.path_planner_name, planner.iter_max))
    out = img_filename or (planner.path_planner_name.replace("*", "star").replace(" ", "_").replace("(", "").replace(")", "").lower() + ".png")
    os.makedirs("visualization/planning_demo", exist_ok=True)
    fig.savefig(os.path.join("visualization/planning_demo", out), dpi=120)
    plt.close(fig)
