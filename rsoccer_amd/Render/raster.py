"""``render_mode="rgb_array"`` without pygame: a small numpy rasteriser.

Stands where rsoccer_gym/Render/{field,robot,ball}.py + the ``render()`` of the base envs stand
(vss_gym_base.py:108-181, ssl_gym_base.py:108-181).  Same window geometry as the reference — VSS:
1.5 x 1.3 m field, 0.1 m margin, 500 px/m -> 750 x 850 x 3; SSL: 9 x 6 m, 0.35 m margin, 100 px/m
-> 670 x 970 x 3 (Render/field.py:189-264) — and the same world -> pixel map (x * scale +
centre, no y flip, vss_gym_base.py:110-113), so code that consumes the reference's frames (video
writers, wrappers) gets arrays of the shape it expects.  ``render_mode="human"`` needs a window
system and stays out of scope.
"""
import numpy as np

BG = (20, 90, 45)
LINE = (255, 255, 255)
BLUE = (0, 64, 255)
YELLOW = (250, 218, 94)
BALL = (253, 106, 2)
MARK = (25, 25, 25)

# length, width, margin, centre-circle radius, penalty length, penalty width, goal width, goal depth, px/m
VSS_VIEW = dict(length=1.5, width=1.3, margin=0.1, circle=0.2, pen_len=0.15, pen_wid=0.7,
                goal_wid=0.4, goal_dep=0.1, scale=500, robot=0.04, ball=0.0215, square=True)
SSL_VIEW = dict(length=9.0, width=6.0, margin=0.35, circle=1.0, pen_len=1.0, pen_wid=2.0,
                goal_wid=1.0, goal_dep=0.18, scale=100, robot=0.09, ball=0.0215, square=False)


class FieldRaster:
    def __init__(self, view):
        self.v = v = dict(view)
        s = v["scale"]
        self.w = int(v["length"] * s + 2 * (v["margin"] * s))   # Render/field.py:27-31 after :34-35,207
        self.h = int(v["width"] * s + 2 * (v["margin"] * s))
        self.cx = (v["length"] / 2 + v["margin"]) * s
        self.cy = (v["width"] / 2 + v["margin"]) * s
        self.window_size = (self.w, self.h)
        self._yy, self._xx = np.mgrid[0:self.h, 0:self.w]
        self._field = self._draw_field()

    # ---- primitives (pixel coordinates) ----
    def _rect(self, img, x0, y0, w, h):
        x0, y0, x1, y1 = int(round(x0)), int(round(y0)), int(round(x0 + w)), int(round(y0 + h))
        x0c, x1c = max(x0, 0), min(x1, self.w - 1)
        y0c, y1c = max(y0, 0), min(y1, self.h - 1)
        for y in (y0, y1):
            if 0 <= y < self.h:
                img[y, x0c:x1c + 1] = LINE
        for x in (x0, x1):
            if 0 <= x < self.w:
                img[y0c:y1c + 1, x] = LINE

    def _disc(self, img, px, py, r, color):
        x0, x1 = max(int(px - r) - 1, 0), min(int(px + r) + 2, self.w)
        y0, y1 = max(int(py - r) - 1, 0), min(int(py + r) + 2, self.h)
        if x0 >= x1 or y0 >= y1:
            return
        m = (self._xx[y0:y1, x0:x1] - px) ** 2 + (self._yy[y0:y1, x0:x1] - py) ** 2 <= r * r
        img[y0:y1, x0:x1][m] = color

    def _line(self, img, xa, ya, xb, yb, color):
        n = int(max(abs(xb - xa), abs(yb - ya))) + 1
        xs = np.clip(np.rint(np.linspace(xa, xb, n)).astype(int), 0, self.w - 1)
        ys = np.clip(np.rint(np.linspace(ya, yb, n)).astype(int), 0, self.h - 1)
        img[ys, xs] = color

    def _draw_field(self):
        v, s = self.v, self.v["scale"]
        img = np.empty((self.h, self.w, 3), np.uint8)
        img[:] = BG
        m, L, W = v["margin"] * s, v["length"] * s, v["width"] * s
        self._rect(img, m, m, L, W)                                           # touch / goal lines
        self._line(img, self.cx, m, self.cx, m + W, LINE)                     # halfway line
        ring = np.abs(np.hypot(self._xx - self.cx, self._yy - self.cy) - v["circle"] * s) <= 0.6
        img[ring] = LINE                                                       # centre circle
        pl, pw = v["pen_len"] * s, v["pen_wid"] * s
        self._rect(img, m, self.cy - pw / 2, pl, pw)                          # penalty areas
        self._rect(img, m + L - pl, self.cy - pw / 2, pl, pw)
        gd, gw = v["goal_dep"] * s, v["goal_wid"] * s
        self._rect(img, m - gd, self.cy - gw / 2, gd, gw)                     # goals
        self._rect(img, m + L, self.cy - gw / 2, gd, gw)
        return img

    # ---- one frame ----
    def draw(self, frame):
        """frame: Entities.Frame (ball + robots_blue / robots_yellow dicts) -> uint8 [H, W, 3]."""
        v, s = self.v, self.v["scale"]
        img = self._field.copy()
        for robots, color in ((frame.robots_blue, BLUE), (frame.robots_yellow, YELLOW)):
            for rb in robots.values():
                px, py = rb.x * s + self.cx, rb.y * s + self.cy
                th = np.deg2rad(rb.theta if rb.theta is not None else 0.0)
                r = v["robot"] * s
                if v["square"]:   # VSS robots are 8 cm cubes: a rotated square
                    c, sn = np.cos(th), np.sin(th)
                    dx, dy = self._xx - px, self._yy - py
                    x0, x1 = max(int(px - 1.5 * r), 0), min(int(px + 1.5 * r) + 1, self.w)
                    y0, y1 = max(int(py - 1.5 * r), 0), min(int(py + 1.5 * r) + 1, self.h)
                    if x0 < x1 and y0 < y1:
                        u = dx[y0:y1, x0:x1] * c + dy[y0:y1, x0:x1] * sn
                        w = -dx[y0:y1, x0:x1] * sn + dy[y0:y1, x0:x1] * c
                        img[y0:y1, x0:x1][(np.abs(u) <= r) & (np.abs(w) <= r)] = color
                else:
                    self._disc(img, px, py, r, color)
                self._line(img, px, py, px + r * np.cos(th), py + r * np.sin(th), MARK)   # heading
        self._disc(img, frame.ball.x * s + self.cx, frame.ball.y * s + self.cy, max(v["ball"] * s, 2.0), BALL)
        return img
