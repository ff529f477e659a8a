"""Palette restatement (magical/style.py:1-40).  Uses the stdlib colorsys like
the reference does; values are float RGB in [0,1]."""
import colorsys


def rgb(r, g, b):
    return (r / 255.0, g / 255.0, b / 255.0)


def darken_rgb(c):  # style.py:10-14
    h, l, s = colorsys.rgb_to_hls(*c)
    return colorsys.hls_to_rgb(h, max(0, l * 0.9), s)


def lighten_rgb(c, times=1):  # style.py:17-22
    h, l, s = colorsys.rgb_to_hls(*c)
    mult = 1.4**times
    return colorsys.hls_to_rgb(h, 1 - (1 - l) / mult, s)


GOAL_LINE_THICKNESS = 0.01
SHAPE_LINE_THICKNESS = 0.015
ROBOT_LINE_THICKNESS = 0.01
COLOURS_RGB = {  # style.py:28-37
    'blue': lighten_rgb(rgb(0x3B, 0x7E, 0xA1), 1.7),
    'yellow': lighten_rgb(rgb(0xFD, 0xB5, 0x15), 1.7),
    'red': lighten_rgb(rgb(0xEE, 0x1F, 0x60), 1.7),
    'green': lighten_rgb(rgb(0x85, 0x94, 0x38), 1.7),
    'grey': rgb(162, 163, 175),
    'brown': rgb(224, 171, 118),
}
ARENA_ZOOM_OUT = 1.02


def to_u8(c):
    """GL float colour -> RGB8 framebuffer (round to nearest)."""
    return tuple(int(v * 255.0 + 0.5) for v in c)
