// STUB (see value.h)
#pragma once
#include "value.h"
