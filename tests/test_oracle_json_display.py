"""oracle/json_display.py against the vectors the reference's own tests hold for `Cell::Json(j) => j.to_string()`
(tests/golden/json_display_kats.py: PINNED), the restated serde_json rules (RESTATED), and the BigQuery integer rule on the PARSED value."""
import pytest

from oracle import json_display as J
from tests.golden import json_display_kats as K


@pytest.mark.parametrize("src,want", K.PINNED)
def test_reference_vectors(src, want):
    assert J.display(src).decode() == want


@pytest.mark.parametrize("src,want", K.RESTATED)
def test_restated_rules(src, want):
    assert J.display(src).decode() == want
    assert J.display(want).decode() == want           # Display is a fixed point of parse + Display


def test_bigquery_rule_walks_the_parsed_value():
    from oracle import protobuf as PB
    from tests.golden import bigquery_kats as BK
    for t in BK.JSON_ACCEPTED + ['{"a":99999999999999999999,"a":1}']:
        PB.validate_json_for_bigquery(t)
    for t in BK.JSON_REFUSED + ['{"a":1,"a":99999999999999999999}']:
        with pytest.raises(PB.UnsupportedValueInDestination):
            PB.validate_json_for_bigquery(t)


def test_device_limits_model():
    assert J.device_limits_ok("[" * 16 + "]" * 16) and not J.device_limits_ok("[" * 17 + "]" * 17)
    assert J.device_limits_ok("{" + ",".join(f'"k{i}":1' for i in range(64)) + "}")
    assert not J.device_limits_ok("{" + ",".join(f'"k":{i}' for i in range(65)) + "}")       # repeated keys count: the text is what a lane scans
    assert not J.device_limits_ok('[{"\\u0024serde_json::private::Number":"1"}]')


def test_json_array_elements():
    from etl_amd import abi
    from oracle import arrays as A
    from oracle.rowbinary import NeedsHost
    assert A.E_JSON == abi.E_JSON
    assert A.elements(199, b'{"{\\"b\\":1,\\"a\\":2}",NULL,"[1, 2]",1e5}') == [(b'{"a":2,"b":1}', None), (None, None), (b"[1,2]", None), (b"1e+5", None)]
    with pytest.raises(A.JsonDecodeError):
        A.elements(3807, b'{"{"}')
    with pytest.raises(NeedsHost):
        A.elements(3807, b'{"' + b"[" * 17 + b"]" * 17 + b'"}')
    with pytest.raises(NeedsHost):
        A.elements(199, b'{"\\"' + b"x" * 260 + b'\\""}')
