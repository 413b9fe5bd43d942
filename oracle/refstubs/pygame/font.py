def init():
    pass


class SysFont(object):
    def __init__(self, *_a, **_k):
        pass

    def render(self, *_a, **_k):
        return None
