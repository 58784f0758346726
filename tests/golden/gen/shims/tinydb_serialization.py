class Serializer:
    pass


class SerializationMiddleware:
    def __init__(self, *a, **k):
        pass

    def register_serializer(self, *a, **k):
        pass
