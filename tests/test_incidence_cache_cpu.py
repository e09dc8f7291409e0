"""The face -> incidence cache of the face_vertices operator (soft_renderer/functional/geometry.py): keyed by the caller's tensor OBJECT,
checked against its storage / shape / dtype / version, built at the second sighting; a full cache drops dead tensors, then tensors whose
structure was never built -- a structure that was handed out stays (it may sit in a captured graph)."""
import torch

from lasr_amd.soft_renderer.functional import geometry as g


def test_incidence_cache_identity_version_and_eviction():
    g._INC_CACHE.clear()
    f = torch.tensor([[[0, 1, 2], [0, 2, 3]]])
    assert g._incidence_of(f, 4) is None                      # first sighting: nothing is built
    inc = g._incidence_of(f, 4)
    assert inc is not None and inc[0].tolist() == [[0, 2, 3, 5, 6]] and inc[1].tolist() == [[0, 3, 1, 2, 4, 5]]
    assert g._incidence_of(f, 4) is inc
    keep = [torch.zeros(1, 1, 3, dtype=torch.long) for _ in range(3 * g._INC_CACHE_MAX)]
    for k in keep:
        g._incidence_of(k, 1)
    # the cache is full of live tensors: f's entry (and the structure it handed out) is still there; late comers take the slots of
    # tensors that were only seen once (least recently seen first)
    assert len(g._INC_CACHE) <= g._INC_CACHE_MAX and g._incidence_of(f, 4) is inc and id(keep[-1]) in g._INC_CACHE and id(keep[0]) not in g._INC_CACHE
    del keep, k
    x = torch.zeros(1, 1, 3, dtype=torch.long)
    g._incidence_of(x, 1)                                     # dead entries make room
    assert id(x) in g._INC_CACHE and len(g._INC_CACHE) <= g._INC_CACHE_MAX
    f[0, 0, 0] = 1                                            # in-place edit: new version, the old structure is not reused
    assert g._incidence_of(f, 4) is None
    inc2 = g._incidence_of(f, 4)
    assert inc2 is not None and inc2 is not inc and inc2[0].tolist() == [[0, 1, 3, 5, 6]]
    assert g._incidence_of(f, 5) is None                      # another vertex count: another structure
    g._INC_CACHE.clear()


def test_incidence_cache_recognises_an_int32_face_tensor_and_can_be_invalidated():
    g._INC_CACHE.clear()
    f32 = torch.tensor([[[0, 1, 2], [0, 2, 3]]], dtype=torch.int32)
    assert g._incidence_of(f32, 4, f32.long()) is None        # the key is the caller's tensor, not its int64 copy (new in every call)
    inc = g._incidence_of(f32, 4, f32.long())
    assert inc is not None and g._incidence_of(f32, 4, f32.long()) is inc
    f32.data[0, 0, 0] = 1                                     # an edit the version counter does not see ...
    assert g._incidence_of(f32, 4) is inc                     # ... is not noticed (documented) ...
    g.invalidate_incidence(f32)                               # ... until the caller says so
    assert g._incidence_of(f32, 4) is None and g._incidence_of(f32, 4)[0].tolist() == [[0, 1, 3, 5, 6]]
    g.invalidate_incidence()
    assert not g._INC_CACHE
