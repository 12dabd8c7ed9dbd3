"""
Genome lookup by identifier and conversion of whatever is found to an animation
(role of cuburn/genome/db.py:6-88): a directory of JSON files, one JSON file holding many
genomes, or a flam3 XML file given directly by path.
"""
import json
import os
import warnings

from . import convert


class GenomeDB(object):
    def __init__(self):
        self.stashed = {}

    def _get(self, id):
        raise NotImplementedError()

    def get(self, id):
        stashed = getattr(self, 'stashed', None) or {}
        return stashed[id] if id in stashed else self._get(id)

    def stash(self, id, gnm):
        if not hasattr(self, 'stashed'):
            self.stashed = {}
        self.stashed[id] = gnm

    def get_anim(self, name, half=False):
        """
        ``name``: genome id, or path of a .flam3/.flame file.  Nodes and edges are converted.
        Returns ``(animation dict, basename for output files)``.
        """
        basename = os.path.basename(name)
        head, dot, ext = basename.rpartition('.')
        if dot and ext in ('json', 'flam3', 'flame'):
            basename = head
        if os.path.isfile(name) and dot and ext in ('flam3', 'flame'):
            with open(name) as fp:
                flames = convert.XMLGenomeParser.parse(fp.read())
            if len(flames) != 1:
                warnings.warn('%d flames in file, only using one.' % len(flames))
            gnm = convert.flam3_to_node(flames[0])
        else:
            gnm = self.get(name)
        if gnm['type'] == 'node':
            gnm = convert.node_to_anim(self, gnm, half=half)
        elif gnm['type'] == 'edge':
            gnm = convert.edge_to_anim(self, gnm)
        assert gnm['type'] == 'animation', 'Unrecognized genome type.'
        return gnm, basename


class OneFileDB(GenomeDB):
    def __init__(self, dct):
        assert dct.get('type') == 'onefiledb', "Doesn't look like a OneFileDB."
        GenomeDB.__init__(self)
        self.dct = dct

    @classmethod
    def read(cls, path):
        with open(path) as fp:
            return cls(json.load(fp))

    def _get(self, id):
        return self.dct[id]


class FilesystemDB(GenomeDB):
    def __init__(self, path):
        GenomeDB.__init__(self)
        self.path = path

    def _get(self, id):
        if not id.endswith('.json'):
            id += '.json'
        with open(os.path.join(self.path, id)) as fp:
            return json.load(fp)


def connect(path):
    return OneFileDB.read(path) if os.path.isfile(path) else FilesystemDB(path)
