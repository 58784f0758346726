class MemoryStorage:
    pass


class JSONStorage:
    pass
