"""Stub: the reference does `from turtle import right` (junk import; tkinter is absent here)."""


def right(*a, **k):
    pass
