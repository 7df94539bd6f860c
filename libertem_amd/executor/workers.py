"""Worker descriptions (subset of the reference's common/scheduler.py:10-154)."""


class Worker:
    def __init__(self, name, host, resources, nthreads):
        self.name = name
        self.host = host
        self.resources = resources
        self.nthreads = nthreads

    def __repr__(self):
        return f"<Worker {self.name} on {self.host} {self.resources}>"


class WorkerSet:
    def __init__(self, workers):
        self.workers = list(workers)

    def __iter__(self):
        return iter(self.workers)

    def __len__(self):
        return len(self.workers)

    def has_cpu(self):
        return WorkerSet([w for w in self.workers if w.resources.get('CPU')])

    def has_hip(self):
        return WorkerSet([w for w in self.workers if w.resources.get('HIP')])

    def hosts(self):
        return {w.host for w in self.workers}
