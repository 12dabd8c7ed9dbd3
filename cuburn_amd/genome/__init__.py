"""Genome format support: the JSON schema and accessors either side of the render boundary."""
