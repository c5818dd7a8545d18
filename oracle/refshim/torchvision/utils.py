def save_image(*a, **k):
    pass


def make_grid(*a, **k):
    raise NotImplementedError
